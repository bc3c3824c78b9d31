"""Why does ICL 00To50_Tracker_ProjectiveBruteforce (tests/test_trackers.cpp:90-162, bound 0.02 m / 0.01) miss its bound under the
`H + lambda I` reading of IterationAlgorithmGN's damping (1.19 x the bound, profiles/r04/sweep_a13_grid.txt column 11) and meet it
under `H + lambda diag(H)` (0.49)?  Arithmetic or scenario?  CPU checker only.

The scenario: frames 00, 01, 50 of ICL lr-kt0 through adaptor -> clipper -> circle finder + depth aligner (icl.conf: lambda 0.1, 100
iterations, inlier-only runs) -> depth-EKF merger.  Between frames 01 and 50 the camera turns by 19 degrees: the constant-velocity
prediction of the motion-model slice is 49 frames off, so frame 50 is the hard step.  Varied here: the damping form, lambda, and what
initialises the aligner's estimate (`use_prediction`: True = the motion-model slice predicts, False = the tracker's last pose).
    python tools/study_icl_tracker.py [--out profiles/r06/icl_tracker_damping_study.txt]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_pins as rp  # noqa: E402
import ref_tracker as rt  # noqa: E402
from oracle import binding as ob  # noqa: E402
from test_ref_pins import OracleBackend  # noqa: E402
from test_ref_tracker import OracleStages  # noqa: E402

B = OracleBackend()
BOUND = np.array((0.02,) * 3 + (0.01,) * 3)


def run(damping_form, damping, use_prediction, max_iterations=None):
    ob.set_variant(damping_form=damping_form)
    try:
        cfg, merger = rt.icl_setup()
        al = dict(cfg["aligner"], damping=damping)
        if max_iterations:
            al["max_iterations"] = max_iterations
        cfg = dict(cfg, aligner=al)
        t = rt.Tracker(OracleStages(), cfg, merger, use_prediction=use_prediction)
        log = [t.process(*rt.icl_measurements(B, k)) for k in (0, 1, 50)]
        err = rp.t2tnq(np.linalg.inv(np.asarray(t.pose, np.float64)) @ rp.icl_relative(50, 0))
    finally:
        ob.set_variant()
    return log, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    lines = [__doc__.strip().split("\n\n")[0], ""]
    lines.append("%-14s %-7s %-16s | %-44s | worst / bound | frame 50: corr  inliers  status" % ("damping", "lambda", "estimate from", "error t2tnq(pose^-1 * truth)"))
    for form, fname in ((0, "lambda diag(H)"), (1, "lambda I")):
        for lam in (0.0, 0.1, 1.0, 10.0):
            for pred, pname in ((True, "motion model"), (False, "last pose")):
                try:
                    log, err = run(form, lam, pred)
                    e50 = log[2]
                    lines.append("%-14s %-7g %-16s | %s | %5.2f | %4d  %4d  %d" % (
                        fname, lam, pname, " ".join("%+.4f" % v for v in err), float(np.max(np.abs(err) / BOUND)), e50.get("n_corr", -1), e50.get("inliers", -1),
                        e50.get("status", -1)))
                except Exception as exc:  # a reading that breaks the loop
                    lines.append("%-14s %-7g %-16s | %s" % (fname, lam, pname, repr(exc)[:80]))
                print(lines[-1], flush=True)
    # which component decides, and how far the two forms end up from each other
    (_, e0), (_, e1) = run(0, 0.1, True), run(1, 0.1, True)
    lines += ["", "shipped (lambda diag(H), 0.1) error / bound per component: " + " ".join("%.2f" % v for v in np.abs(e0) / BOUND),
              "lambda I, 0.1                  error / bound per component: " + " ".join("%.2f" % v for v in np.abs(e1) / BOUND),
              "difference of the two final poses' errors [m, quaternion units]: " + " ".join("%+.4f" % v for v in (e1 - e0))]
    for ln in lines[-3:]:
        print(ln)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

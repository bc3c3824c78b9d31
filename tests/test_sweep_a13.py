"""The round-4 decision on the BUILD-DEFINED arithmetic of rows a13 / a14 as a gating assertion: under the shipped family EVERY pose
bound the reference's gtests assert on its own images holds (tests/test_aligners.cpp:586-1340, tests/test_trackers.cpp:7-783: 17
assertions), and the readings rounds 2-3 shipped do not.  The full table (128 families) is profiles/r04/sweep_a13_grid.txt,
written by tools/sweep_a13.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def sweep():
    import sweep_a13
    from oracle import binding as ob
    yield sweep_a13
    ob.set_variant()  # back to the shipped definition whatever happened


def _cells(sweep, guess, **variant):
    from oracle import binding as ob
    ob.set_variant(**variant)
    try:
        out = {}
        for name, fn in sweep.scenarios(False, guess).items():
            err, bound = fn()
            out[name] = sweep.ratio(err, bound)
        return out
    finally:
        ob.set_variant()


def test_shipped_family_meets_every_reference_pose_bound(sweep):
    cells = _cells(sweep, 1)
    assert len(cells) == 17
    thin = {k: round(v, 3) for k, v in sorted(cells.items(), key=lambda kv: -kv[1]) if v >= 0.9}
    # margin-aware failure message: the thin cells (>= 0.9 of their bound; the KITTI tracker's forward axis sits at 0.90 - 0.99) are
    # named with their values, so that a libm / compiler change that tips one over is diagnosable from the log alone
    assert all(v < 1.0 for v in cells.values()), "cells at or above their bound: %s; all thin cells: %s" % (
        {k: round(v, 3) for k, v in cells.items() if v >= 1.0}, thin)
    if thin:
        import warnings
        warnings.warn("a13 family: %d of 17 reference pose bounds are met with less than 10 %% margin: %s" % (len(thin), thin))


def test_the_readings_of_rounds_2_and_3_miss_a_tracker_bound(sweep):
    # kernelised factors weighted tau / chi, H + lambda I, clamp(d / mean, 0.01, 1), start = the caller's estimate
    cells = _cells(sweep, 0, kernel_form=1, idw_form=1, damping_form=1)
    missed = [k for k, v in cells.items() if v >= 1.0]
    assert any(k.startswith("T:nomerge") for k in missed), missed

"""prints the figures of a bench.py JSON line (file argument) in a few lines"""
import json
import sys

for f in sys.argv[1:]:
    txt = [l for l in open(f) if l.startswith("{")]
    if not txt:
        print(f, "EMPTY")
        continue
    d = json.loads(txt[-1])
    print(f, "value %.0f n_gpus %d ms/step %.2f" % (d["value"], d["n_gpus"], d["ms_per_step"] or 0), d.get("ranks", ""))
    for k in ("roofline_search", "roofline_gn", "roofline_matcher"):
        r = d.get(k)
        if r:
            print("  %-17s ms %.3f frac %.3f traffic/alg %s rounds %s" % (k, r.get("ms_per_step", r.get("ms_per_launch")), r["frac"], r.get("traffic_over_algorithmic"),
                                                                          [round(x, 2) for x in r.get("ms_by_round", [])]))
    print("  cfg", {k: v for k, v in d["config"].items() if k not in ("workload", "parallelism")})
    if "steady_state" in d:
        print("  steady %.0f" % d["steady_state"]["value"], d["steady_state"]["ms_per_kernel"])
    for n, o in d.get("other_configs", {}).items():
        print("   %-12s %.0f" % (n, o.get("value", 0)), {k: round(v, 2) for k, v in o.get("ms_per_kernel", {}).items()}, o.get("parity"), o.get("gn_iterations_executed_mean"))
    for k in ("closed_loop", "kitti_real", "from_images", "latency_b1"):
        if k in d:
            c = d[k]
            print("  %s" % k, {kk: vv for kk, vv in c.items() if kk not in ("config", "cpu_baseline")})
    print("  cpu", d.get("cpu_baseline"), d.get("cpu_baseline_all_cores"), d.get("parity_on_bench_inputs"))

"""Config 5 on N GPUs, simulated from a measured 1-GPU run of the whole batch (bench.py --config 5 --farm-record <json>):
how long would 1 / 2 / 4 / 8 ranks of `lanes` slice contexts each need, with the round robin slice i -> rank i % N that
farm.py had until round 4 ("static") and with the shared queue it has now, in index order and longest first (by the measured
iteration counts: "when iteration counts are known")?  List scheduling with the measured per-slice milliseconds -- which were
taken with `lanes` contexts sharing one GPU, exactly what every simulated rank runs.  No 8-GPU node was available to this
build: this is the evidence VERDICT r4 item 2 asks for, not a measurement.

    python scripts/makespan_sim.py profiles/r6_config5_512slices_1gpu.json [profiles/r6_config5_makespan.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import farm


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r6_config5_makespan.json")
    rec = json.loads(open(src).read().strip().splitlines()[-1])
    per = rec["config"]["per_slice"]
    ms, its = per["solve_ms"], per["iterations"]   # (a lane's own time per slice; "ms" also counts the wait under its previous solve)
    lanes = int(rec["config"]["parallelism"].split(" x ")[1].split()[0])
    measured_s = rec["ms_per_step"] / 1e3
    base = farm.simulate_makespan(ms, 1, lanes) / 1e3
    out = {
        "source": os.path.basename(src), "slices": len(ms), "lanes_per_rank": lanes,
        "measured_1gpu_s": measured_s, "simulated_1gpu_s": base,
        "iterations": {"mean": sum(its) / len(its), "max": max(its), "min": min(its)},
        "slice_ms": {"mean": sum(ms) / len(ms), "max": max(ms), "sum_s": sum(ms) / 1e3},
        "note": "list scheduling of the measured per-slice milliseconds (taken with %d contexts sharing one GPU) over ranks x lanes; "
                "speed-up = simulated 1-GPU makespan / simulated N-GPU makespan; simulated, not measured" % lanes,
        "ranks": {},
    }
    for world in (1, 2, 4, 8):
        st = farm.simulate_makespan(ms, world, lanes, static=True) / 1e3
        dy = farm.simulate_makespan(ms, world, lanes) / 1e3
        lp = farm.simulate_makespan(ms, world, lanes, costs=its) / 1e3
        out["ranks"][str(world)] = {
            "static_round_robin": {"makespan_s": st, "speedup": base / st},
            "shared_queue_index_order": {"makespan_s": dy, "speedup": base / dy},
            "shared_queue_longest_first": {"makespan_s": lp, "speedup": base / lp},
            "ideal_s": sum(ms) / 1e3 / (world * lanes),
        }
    json.dump(out, open(dst, "w"), indent=1)
    for w, r in out["ranks"].items():
        print("%s rank(s): static %.2f s (x%.2f)  queue %.2f s (x%.2f)  queue, longest first %.2f s (x%.2f)  ideal %.2f s" %
              (w, r["static_round_robin"]["makespan_s"], r["static_round_robin"]["speedup"], r["shared_queue_index_order"]["makespan_s"],
               r["shared_queue_index_order"]["speedup"], r["shared_queue_longest_first"]["makespan_s"], r["shared_queue_longest_first"]["speedup"], r["ideal_s"]))


if __name__ == "__main__":
    main()

"""Per-rank device time of an LM iteration from bench.py lines taken with the one-GPU hook (ranks take turns: uncontended times).
usage: python tools/hook_summary.py gpurun_out/bench_hook_c4_n8.json ..."""
import json, sys
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    lm = d["lm"]
    if "per_rank" not in lm:
        print(path, "no per_rank:", lm.get("error")); continue
    rows = lm["per_rank"]
    n = d["n_gpus"]
    # ("other" — two tiny launches — is the first thing a rank runs when its turn on the shared GPU begins: it absorbs the switch between the processes)
    keys = [k for k in rows[0]["phase_ms_per_lm_iteration"] if k not in ("exchange", "other")]
    worst = {k: max(r["phase_ms_per_lm_iteration"].get(k, 0.0) for r in rows) for k in keys}
    tot = max(sum(r["phase_ms_per_lm_iteration"].get(k, 0.0) for k in keys) for r in rows)
    col = rows[0]["collectives_of_the_profiled_solve"]
    iters = max(1, lm["iterations"])
    print(f"{d['config']['workload'].split(':')[0]} N={n}: slowest rank's device time per LM iteration without the collectives (and without 'other') {tot:.3f} ms; plan {rows[0]['plan']}")
    print("   per phase (max over ranks): " + "  ".join(f"{k} {v:.3f}" for k, v in worst.items()))
    print("   collectives per iteration: " + "; ".join(f"{k}: {v['calls'] / iters:.1f} calls, {v['bytes'] / iters / 1e6:.3f} MB" for k, v in col.items() if v["calls"]))

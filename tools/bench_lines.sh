# the three bench lines of a round in a call of their own (after the PMC summaries of the same build are under profiles/$1)
export TMPDIR=/tmp
R=${1:-r06}; O=gpurun_out/$R; mkdir -p $O
python bench.py --steps 50 --warmup 5 --lm-iters 12 > $O/bench.json 2> $O/bench.err
python bench.py --config C5 --steps 20 --warmup 3 --lm-iters 8 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config C2 --steps 50 --warmup 5 --lm-iters 12 > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
for f in ("bench.json", "bench_c5.json", "bench_c2.json"):
    d = json.load(open("$O/" + f)); r = d["roofline"]; tp = r.get("traffic_provenance") or {}
    print(f, "value %.4g" % d["value"], "frac %.3f" % r["frac"], tp.get("file"), "stale", tp.get("stale"), "lm", round(d["lm_headline"]["ms_per_lm_iteration"], 4), "first_solve", round(d["lm_headline"]["first_solve_wall_s"], 4))
PY

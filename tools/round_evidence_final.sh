# final evidence run of a round: tools/round_evidence.sh, then the PMC summaries go where bench.py looks for them (profiles/$R on the box's copy of
# the tree) and the three bench lines are taken again, so that their traffic / mfma_busy provenance is this very build's
export TMPDIR=/tmp
R=${1:-r06}; O=gpurun_out/$R
bash tools/round_evidence.sh $R > gpurun_out/$R.round3.log 2>&1
mkdir -p profiles/$R
cp $O/pmc_summary.json $O/pmc_mfma_summary.json $O/pmc_valu_summary.json profiles/$R/
python bench.py --steps 50 --warmup 5 --lm-iters 12 > $O/bench.json 2> $O/bench.err
python bench.py --config C5 --steps 20 --warmup 3 --lm-iters 8 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config C2 --steps 50 --warmup 5 --lm-iters 12 > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json
for f in ("bench.json", "bench_c5.json", "bench_c2.json"):
    d = json.load(open("$O/" + f)); r = d["roofline"]
    tp = r.get("traffic_provenance") or {}
    print(f, "value %.4g" % d["value"], "frac %.3f" % r["frac"], tp.get("file"), "stale", tp.get("stale"), "lm marginal", round(d["lm_headline"]["marginal_ms_per_lm_iteration"], 4))
PY
tail -3 $O/pytest_gpu.txt

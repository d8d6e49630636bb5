#!/bin/bash
# Multi-process stress of the persistent Cholesky driver: N processes share the GPU, each solving random problems with the
# DAG driver and with the level schedule and comparing bit for bit (tools/dag_stress.py).  usage: stress_multi.sh SECONDS N OUTDIR [seed0]
SECS=${1:-300}; N=${2:-6}; OUT=${3:-gpurun_out/stress}; S0=${4:-100}
mkdir -p $OUT
for k in $(seq 1 $N); do
  python tools/dag_stress.py $SECS $((S0 + k)) > $OUT/stress_$((S0 + k)).log 2>&1 &
done
wait
grep -h -E "dag_stress|MISMATCH" $OUT/stress_*.log | tee $OUT/summary_$S0.txt
python - $OUT <<'PY'
import glob, re, sys
tot = [0, 0, 0]
for f in glob.glob(sys.argv[1] + "/stress_*.log"):
    last = None
    for line in open(f):
        m = re.search(r"(\d+) (?:random )?problems, (\d+) mismatches, (\d+) (?:verification )?fallbacks", line)
        if m: last = m
    if last: tot = [a + int(b) for a, b in zip(tot, last.groups())]
print(f"TOTAL {tot[0]} problems (each solved with both drivers), {tot[1]} mismatches, {tot[2]} verification fallbacks")
PY

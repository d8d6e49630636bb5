#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass (counters only: never combined with tracing) over a command, summarised per kernel
# (mean per dispatch) into OUT.csv.   usage: tools/pmc_pass.sh OUT "COUNTER1 COUNTER2 .." command...
OUT=$1; CNT=$2; shift 2
export TMPDIR=/tmp
D=$(mktemp -d /tmp/pmc.XXXXXX)
rocprofv3 --pmc $CNT --output-format csv -d $D -o p -- "$@" > $OUT.log 2>&1
python - "$D" "$OUT.csv" <<'PY'
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(sys.argv[2], "w") as g:
    g.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -max(v[1] for v in acc[k].values())):
        n = max(v[1] for v in acc[k].values())
        g.write('"%s",%d,' % (k, n) + ",".join("%.6g" % (acc[k][c][0] / max(1, acc[k][c][1])) for c in names) + "\n")
PY
rm -rf $D

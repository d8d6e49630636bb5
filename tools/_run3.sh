cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x3
RSBA_DEVICE_LM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x3/tr -o t -- python tools/gap_probe.py C2 24 > gpurun_out/x3/gap.log 2>&1
f=$(find gpurun_out/x3/tr -name "t_kernel_trace.csv" | head -1)
python tools/gap_probe.py --read $f 24 list > gpurun_out/x3/list_c2.txt 2>&1
rm -rf gpurun_out/x3/tr
cat gpurun_out/x3/list_c2.txt

# final-code pass: stress of the persistent driver, then the bench + kernel stats + PMC passes again (their stamp must be the final code's)
export TMPDIR=/tmp
R=${1:-r05}; O=gpurun_out/$R; mkdir -p $O
bash tools/stress_multi.sh 240 6 $O/stress 100 > $O/dag_stress_summary.txt 2>&1
bash tools/profile_round.sh $R > gpurun_out/$R.round.log 2>&1
{ RSBA_DEBUG_PLAN=1 python tools/setup_time.py C4 2>&1 | grep -v amdgpu; python tools/setup_time.py C2 2>&1 | grep -v amdgpu; RSBA_DEBUG_PLAN=1 python tools/setup_time.py C5 2>&1 | grep -E "host phases|fresh handle|device lists"; } > $O/setup_time.txt
python tools/chol_trace.py C4 2>&1 | grep -v amdgpu > $O/chol_trace_c4.txt
tail -3 $O/dag_stress_summary.txt; tail -c 600 $O/bench.json; cat $O/pmc_summary.json | tail -4

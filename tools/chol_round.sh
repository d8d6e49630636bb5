#!/bin/bash
# Runs on the GPU box: the Cholesky work loop — solver parity tests, LM wall time per configuration (DAG and level schedule,
# bit-compared), the task trace of the DAG driver.  usage: chol_round.sh OUTDIR [quick]
OUT=${1:-gpurun_out/chol}; mkdir -p $OUT
timeout 120 ./tools/tile_factor_bench > $OUT/tile_factor.txt 2>&1
timeout 200 python tools/lm_time.py C2 6 > $OUT/lm_c2.log 2>&1 || { echo 'C2 solve failed or hung:'; tail -5 $OUT/lm_c2.log; exit 1; }
(timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_priors.py tests/test_gpu_pose_priors.py tests/test_gpu_covariance.py -m gpu -x -q 2>&1 | tail -40) > $OUT/pytest.log
timeout 300 python tools/lm_time.py C4 12 > $OUT/lm_c4.log 2>&1
RSBA_CHOL_FUSE=0 timeout 300 python tools/lm_time.py C4 12 > $OUT/lm_c4_nofuse.log 2>&1
timeout 300 python tools/chol_trace.py C4 > $OUT/chol_trace_c4.txt 2>&1
if [ "$2" != "quick" ]; then
  timeout 300 python tools/lm_time.py C5 8 > $OUT/lm_c5.log 2>&1
  (timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5) > $OUT/pytest_fullsize.log
fi
cat $OUT/tile_factor.txt; tail -30 $OUT/pytest.log | cut -c1-200; cat $OUT/lm_c4.log $OUT/lm_c4_nofuse.log; head -3 $OUT/chol_trace_c4.txt; tail -42 $OUT/chol_trace_c4.txt

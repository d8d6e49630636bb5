#!/bin/bash
# Runs on the GPU box: the Cholesky work loop — solver parity tests, LM wall time per configuration (DAG and level schedule,
# bit-compared), the task trace of the DAG driver.  usage: chol_round.sh OUTDIR [quick]
OUT=${1:-gpurun_out/chol}; mkdir -p $OUT
./tools/tile_factor_bench > $OUT/tile_factor.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_priors.py tests/test_gpu_pose_priors.py tests/test_gpu_covariance.py -m gpu -x -q 2>&1 | tail -5) > $OUT/pytest.log
python tools/lm_time.py C4 12 > $OUT/lm_c4.log 2>&1
RSBA_CHOL_FUSE=0 python tools/lm_time.py C4 12 > $OUT/lm_c4_nofuse.log 2>&1
python tools/chol_trace.py C4 > $OUT/chol_trace_c4.txt 2>&1
if [ "$2" != "quick" ]; then
  python tools/lm_time.py C5 8 > $OUT/lm_c5.log 2>&1
  (timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5) > $OUT/pytest_fullsize.log
fi
cat $OUT/tile_factor.txt; tail -3 $OUT/pytest.log; cat $OUT/lm_c4.log $OUT/lm_c4_nofuse.log; head -12 $OUT/chol_trace_c4.txt

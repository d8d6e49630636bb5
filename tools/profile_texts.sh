#!/bin/bash
# Runs on the GPU box (via gpurun) after tools/profile_round.sh: the text evidence profiles/rNN/README.md lists — phase times, LM
# wall times (DAG driver against the level schedule), the task timeline of the Cholesky, the chunk timeline of the Schur kernel,
# the tile-factorisation and hand-off micro-benchmarks, the cost of a fresh handle, and the -m gpu suite — into gpurun_out/$1/.
R=${1:-r06}
OUT=gpurun_out/$R
mkdir -p $OUT
F='grep -v amdgpu.ids'
{ for C in C4 C5 C2; do IT=12; [ $C = C5 ] && IT=8; python tools/phase_time.py $C $IT 2>&1 | $F; done; python tools/fixed_cost.py 2>&1 | $F; } > $OUT/phase_times.txt
python tools/lm_time.py C4 12 2>&1 | $F > $OUT/lm_time_c4.txt
python tools/lm_time.py C5 8 2>&1 | $F > $OUT/lm_time_c5.txt
python tools/lm_time.py C4 12 free_ratio 2>&1 | $F > $OUT/lm_time_c4_free_ratio.txt
python tools/chol_trace.py C4 2>&1 | $F > $OUT/chol_trace_c4.txt
python tools/schur_trace.py C4 2>&1 | $F > $OUT/schur_trace_c4.txt
timeout 120 tools/tile_factor_bench > $OUT/tile_factor.txt 2>&1
timeout 120 tools/xcd_handoff > $OUT/xcd_handoff.txt 2>&1
{ RSBA_DEBUG_PLAN=1 python tools/setup_time.py C4 2>&1 | $F; python tools/setup_time.py C2 2>&1 | $F; RSBA_DEBUG_PLAN=1 python tools/setup_time.py C5 2>&1 | $F | grep -E "host phases|fresh handle|device lists"; } > $OUT/setup_time.txt
{ for C in C4 C5; do for FAC in 0 1; do echo "== $C RSBA_FACTORED=$FAC"; RSBA_FACTORED=$FAC python tools/phase_time.py $C 8 2>&1 | $F; done; done; } > $OUT/factored_groups_ab.txt
python tools/lm_time.py C4 12 priors 2>&1 | $F > $OUT/lm_time_c4_priors.txt
python tools/filter_time.py 2>&1 | $F > $OUT/filter_time.txt
# problems that keep records (per-frame intrinsics blocks): the device-side loop with a second set of records against round 5's form (one set, host decides)
{ for C in C2 C4; do echo "== $C, a candidate's records in a second set (default)"; python tools/lm_time.py $C 12 per_frame_intrinsics 2>&1 | $F | grep "dag:"; echo "== $C, RSBA_RECORDS_ALT=0 (round 5)"; RSBA_RECORDS_ALT=0 python tools/lm_time.py $C 12 per_frame_intrinsics 2>&1 | $F | grep "dag:"; done; } > $OUT/lm_time_per_frame_intrinsics.txt
{ RSBA_AMD_LIB=rsba_amd/_lib/librsba_amd_hooks.so python tools/chol_poll_split.py C4 8; RSBA_AMD_LIB=rsba_amd/_lib/librsba_amd_hooks.so python tools/chol_poll_split.py C5 4; } 2>&1 | $F > $OUT/chol_poll_split_final.txt
python -m pytest tests -m gpu -q 2>&1 | $F | grep -E "passed|failed|error|Error" | tail -8 > $OUT/pytest_gpu.txt

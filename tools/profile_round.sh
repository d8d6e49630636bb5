#!/bin/bash
# Runs on the GPU box (via gpurun): collects this round's evidence into gpurun_out/$1/ (default r02).
#   the bench lines (C4 = the metric's workload, C5 = the 4k-camera Huber + shared-intrinsics scene), kernel-trace stats of the
#   same commands, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never combined with tracing), the HBM stream calibration.
R=${1:-r06}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 50 --warmup 5 --lm-iters 12 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --config C5 --steps 20 --warmup 3 --lm-iters 8 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
for C in C4 C5; do
  B="python bench.py --config $C --steps 30 --warmup 5 --no-cpu-baseline --no-next-rows --lm-iters 12"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$C -o t -- $B > $OUT/trace_$C.log 2>&1
  cp $OUT/trace_$C/t_kernel_stats.csv $OUT/kernel_stats_$C.csv; rm -rf $OUT/trace_$C
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$C -o f -- $B --no-lm > $OUT/pmc_fetch_$C.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$C -o w -- $B --no-lm > $OUT/pmc_write_$C.log 2>&1
done
# matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES) and L2 behaviour of the LM kernels, counters only (tools/pmc_pass.sh: one --pmc pass each)
for C in C4 C5; do
  IT=4; [ $C = C5 ] && IT=3
  tools/pmc_pass.sh $OUT/pmc_mfma_$C "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python tools/lm_time.py $C $IT
  tools/pmc_pass.sh $OUT/pmc_l2_$C "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" python tools/lm_time.py $C $IT
done
python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
summary = {"note": "per kernel, mean per dispatch: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); one fp64 MFMA 16x16x4 keeps its SIMD's "
                   "matrix pipe busy for 64 cycles (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA); SQ_WAVE_CYCLES, SQ_WAIT_* in quad-cycles; l2_hit = TCC_HIT / TCC_REQ, "
                   "fabric_read_bytes = TCC_EA0_RDREQ x 128 B (Infinity-Cache hits included)"}
for cfg in ("C4", "C5"):
    rows = {r["kernel"]: r for r in csv.DictReader(open(f"{out}/pmc_mfma_{cfg}.csv"))}
    l2 = {r["kernel"]: r for r in csv.DictReader(open(f"{out}/pmc_l2_{cfg}.csv"))}
    d = {}
    for k, r in rows.items():
        gui = float(r["GRBM_GUI_ACTIVE"]) / 8.0
        if gui <= 0 or not k.startswith("rsba::"): continue
        e = {"dispatches": int(r["dispatches"]), "mfma_busy_cycles": float(r["SQ_VALU_MFMA_BUSY_CYCLES"]), "mfma_insts": float(r["SQ_INSTS_MFMA"]), "cycles_per_xcd": gui,
             "mfma_busy_frac": float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui * 1024.0), "waves_per_simd": 4.0 * float(r["SQ_WAVE_CYCLES"]) / (gui * 1024.0),
             "wave_time_waiting_for_memory_or_barriers": float(r["SQ_WAIT_ANY"]) / max(1.0, float(r["SQ_WAVE_CYCLES"]))}
        if k in l2 and float(l2[k]["TCC_REQ_sum"]) > 0:
            e["l2_hit"] = float(l2[k]["TCC_HIT_sum"]) / float(l2[k]["TCC_REQ_sum"]); e["fabric_read_bytes"] = float(l2[k]["TCC_EA0_RDREQ_sum"]) * 128.0
        d[k] = e
    summary[cfg] = d
json.dump(summary, open(f"{out}/pmc_mfma_summary.json", "w"), indent=1)
for cfg in ("C4", "C5"):
    for k, e in summary[cfg].items():
        if e["mfma_busy_frac"] > 0.01: print(cfg, k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()})
PY
# the vector unit of the passes that RECOMPUTE the observation model (eval_trial / point_blocks / project / back_substitute): issued and
# active VALU cycles, and the fp64 operations by kind (a wave instruction = 64 lanes; an FMA counts two flops) — counters only, one pass each
for C in C4 C5; do
  IT=4; [ $C = C5 ] && IT=3
  tools/pmc_pass.sh $OUT/pmc_valu_$C "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" python tools/lm_time.py $C $IT
  tools/pmc_pass.sh $OUT/pmc_f64_$C "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" python tools/lm_time.py $C $IT
  tools/pmc_pass.sh $OUT/pmc_lds_$C "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" python tools/lm_time.py $C $IT
done
python tools/pmc_valu_summary.py $OUT
python bench.py --config C2 --steps 50 --warmup 5 --lm-iters 12 --no-cpu-baseline --no-next-rows > $OUT/bench_c2.json 2> $OUT/bench_c2.err
./tools/hbm_calib > $OUT/hbm_calib.txt 2>&1
./tools/mfma_f64_rate > $OUT/mfma_f64_rate.txt 2>&1
python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
def mean_counter(path, counter, kernel_sub):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
summary = {"note": "gfx950 calibration (tools/hbm_calib.hip under the same counters): FETCH_SIZE reads 0.500x of coalesced read streams, WRITE_SIZE 1.000x; "
                   "hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB, separate --pmc passes"}
for cfg, k, key in (("C4", "eval_kernel<true, 2, 1>", "hbm_bytes_per_launch"), ("C5", "eval_kernel<false, 2, 1>", "hbm_bytes_per_launch_C5")):
    f, nf = mean_counter(f"{out}/pmc_fetch_{cfg}/f_counter_collection.csv", "FETCH_SIZE", k)
    w, nw = mean_counter(f"{out}/pmc_write_{cfg}/w_counter_collection.csv", "WRITE_SIZE", k)
    stats = {r["Name"]: r for r in csv.DictReader(open(f"{out}/kernel_stats_{cfg}.csv"))}
    avg_ns = next((float(v["AverageNs"]) for n, v in stats.items() if k in n), None)
    summary[key] = (2.0 * f + w) * 1024.0 if f and w else None
    summary[cfg] = {"kernel": "rsba::" + k, "launches_fetch": nf, "launches_write": nw, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "rocprof_avg_kernel_ns": avg_ns}
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary))
PY
python tools/source_stamp.py --tag $OUT/pmc_summary.json $OUT/pmc_mfma_summary.json $OUT/pmc_valu_summary.json   # the code the counters were taken on (bench.py flags a stale summary)
cp $OUT/kernel_stats_C4.csv $OUT/kernel_stats.csv
cat $OUT/bench.json
# the rows next to the hot path: motion priors (f1), RS-PnP hypotheses (f3), filters (f2)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_priors -o t -- python tools/lm_time.py C4 12 priors > $OUT/lm_priors.log 2>&1
cp $OUT/trace_priors/t_kernel_stats.csv $OUT/lm_c4_priors_kernel_stats.csv; rm -rf $OUT/trace_priors
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pnp -o t -- python tools/pnp_time.py 16384 1000 6 > $OUT/pnp.log 2>&1
cp $OUT/trace_pnp/t_kernel_stats.csv $OUT/f3_pnp_kernel_stats.csv; rm -rf $OUT/trace_pnp

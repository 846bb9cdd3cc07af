#!/bin/bash
# profiles/collect.sh TAG -- run on the GPU box (gpurun -- 'bash profiles/collect.sh r01c'): rocprofv3 kernel-trace + stats
# and the PMC passes (counters in their own runs) over the default bench.py command, then the text/json summaries that are
# committed under profiles/.  Everything is written under gpurun_out/ (scratch); copy the summaries into profiles/ afterwards.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
BENCH="$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
    --kernel-trace -d $OUT/pmc_sq1 -- python3 $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU \
    --kernel-trace -d $OUT/pmc_sq2 -- python3 $BENCH > $OUT/pmc_sq2.log 2>&1
# the fp32-tensor workload (c3) through the same bench: kernel durations and matrix-pipe / HBM counters of the split kernel
BENCH3="$R/bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/stats_c3 -- python3 $BENCH3 > $OUT/stats_c3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
    --kernel-trace -d $OUT/pmc_c3_sq -- python3 $BENCH3 > $OUT/pmc_c3_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_c3_fetch -- python3 $BENCH3 > $OUT/pmc_c3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_c3_write -- python3 $BENCH3 > $OUT/pmc_c3_write.log 2>&1
# the accurate bf16 path (fp32 output: P as bf16 hi + bf16 lo, ONE launch per forward): durations and matrix-pipe counters
BENCHA="$R/bench.py --accurate --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/stats_acc -- python3 $BENCHA > $OUT/stats_acc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
    --kernel-trace -d $OUT/pmc_acc_sq -- python3 $BENCHA > $OUT/pmc_acc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_acc_fetch -- python3 $BENCHA > $OUT/pmc_acc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_acc_write -- python3 $BENCHA > $OUT/pmc_acc_write.log 2>&1
# shapes that had no profile before round 3: BASELINE config 2 (fp32 tensors, N = 1024; AUTO and exact arithmetic) and the README's d = 32 rows,
# through the C driver (one kernel family per pass; 200 launches each)
DRV=$R/flashattention.c_amd/fa_driver
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
shape_pass() {   # name, driver args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d $OUT/stats_$name -- $DRV --mode rand --check 0 --warmup 50 --iters 200 "$@" > $OUT/stats_$name.log 2>&1
    rocprofv3 --pmc $SQ --kernel-trace -d $OUT/pmc_${name}_sq -- $DRV --mode rand --check 0 --warmup 50 --iters 200 "$@" > $OUT/pmc_${name}_sq.log 2>&1
}
shape_pass c2_auto  --bh 128 --n 1024 --d 64 --dtype f32s --kernel auto
shape_pass c2_exact --bh 128 --n 1024 --d 64 --dtype f32
shape_pass d32_n1024_bf16 --bh 128 --n 1024 --d 32 --dtype bf16
shape_pass d32_n8192_bf16 --bh 16 --n 8192 --d 32 --dtype bf16
shape_pass c4_causal --bh 16 --n 8192 --d 64 --dtype bf16 --causal 1
# round 4: the accurate path (two bf16 terms of P) beyond c4 -- causal, d = 128
shape_pass acc_causal --bh 16 --n 8192 --d 64 --dtype bf16 --kernel pb2 --out_f32 1 --causal 1
shape_pass acc_d128 --bh 16 --n 8192 --d 128 --dtype bf16 --kernel pb2 --out_f32 1
cd $R
# first summary: writes profiles/pmc_traffic.json (with this library's sha256) from the PMC passes above, so that the bench lines below --
# which print `traffic` only for the library the counters were collected on -- carry it; the second one at the end checks the kernel names
python3 profiles/summarize_rocpd.py $OUT $TAG --out $OUT > /dev/null 2>&1 || true
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null || true   # (this box's copy of the tree: what bench.py reads)
python3 bench.py --accurate --no-cpu-baseline --no-extras > $OUT/bench_line_accurate.json 2> $OUT/bench_acc.err
python3 bench.py --workload c3 --no-cpu-baseline > $OUT/bench_line_c3.json 2> $OUT/bench_c3.err
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python3 profiles/summarize_rocpd.py $OUT $TAG --out $OUT || echo "summarize_rocpd.py FAILED (kernel name mismatch?)"
# the raw rocpd databases are tens of MB per pass (gpurun merges at most 64 MiB back): keep the summaries and the logs only
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
grep -h '"metric"' $OUT/*.log | head -3
cat $OUT/bench_line.json

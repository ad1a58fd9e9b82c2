#!/bin/bash
# usage: tools/gpu_profile.sh <tag> [bench args]  -- rocprofv3 kernel stats + PMC passes of bench.py
# trace      : the default command (band-limited rows overlap the two-pass chain on a side stream in the timed loop)
# trace_ser  : the same with --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0 (every kernel alone: durations comparable with bench.py's HIP events)
# pmc_*      : counter passes (kernel trace only, one counter group per pass), serialized kernels
export TMPDIR=/tmp
tag=$1; shift
OUT=$PWD/gpurun_out/$tag
mkdir -p $OUT
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic $*"
SER="$BENCH --opt overlap_narrow=0 --opt ols_early=0 --opt ols_side=0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cwt -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ser -o cwt -- $SER > $OUT/trace_ser.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o cwt -- $SER > $OUT/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq2 -o cwt -- $SER > $OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o cwt -- $SER > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o cwt -- $SER > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o cwt -- $SER > $OUT/pmc_tcc.log 2>&1
python tools/summarize_prof.py $OUT --traffic-json $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace_ser -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_serialized.csv \;
# keep only small files
find $OUT -type f -size +8M -delete

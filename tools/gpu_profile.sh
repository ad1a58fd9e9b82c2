#!/bin/bash
# usage: tools/gpu_profile.sh <tag> [bench args]  -- rocprofv3 kernel stats + PMC passes of bench.py
export TMPDIR=/tmp
tag=$1; shift
OUT=$PWD/gpurun_out/$tag
mkdir -p $OUT
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cwt -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o cwt -- $BENCH > $OUT/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq2 -o cwt -- $BENCH > $OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o cwt -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o cwt -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o cwt -- $BENCH > $OUT/pmc_tcc.log 2>&1
find $OUT -name "*.csv" | head -30
python tools/summarize_prof.py $OUT --traffic-json $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files
find $OUT -type f -size +8M -delete

#!/bin/bash
# Round profile refresh (run on the GPU box through gpurun from the repository root):
#   kernel-trace stats of bench.py (40k columns), then separate PMC passes (never combined with other trace domains).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --variants 40000 --steps 3 --warmup 1 --cpu-baseline-columns 0"
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
SMALL="python $REPO/bench.py --variants 8000 --steps 1 --warmup 1 --cpu-baseline-columns 0"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- $SMALL > $OUT/pmc_$name.log 2>&1
  echo "pmc $grp rc=$?"
done
TRIO="python $REPO/bench.py --trio --variants 20000 --coverage 15 --steps 3 --warmup 1 --cpu-baseline-columns 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_trio -- $TRIO > $OUT/trace_trio.log 2>&1
find $OUT -name "*.csv" | head -40
# keep the merge small: drop the raw kernel traces of the big run, keep stats and counter files
find $OUT/trace $OUT/trace_trio -name "*kernel_trace.csv" -size +4M -delete
du -sh $OUT

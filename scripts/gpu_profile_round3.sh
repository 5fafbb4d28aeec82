#!/bin/bash
# Round-3 profile refresh (run on the GPU box through gpurun from the repository root, stdin closed):
#   1. the default bench line (headline configs[2] + the `configs` array; its own rocprofv3 --pmc passes run inside bench.py)
#   2. rocprofv3 --kernel-trace --stats of the headline, of configs[3] (pedigree slot runs), of the genotyping and heuristic rows
# Counter passes are never combined with other trace domains (bench.py: --kernel-trace --pmc only).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03/final
rm -rf $OUT; mkdir -p $OUT
S=$(date +%s)
timeout 600 python $REPO/bench.py --pmc-keep $OUT/pmc > $OUT/bench_default.json 2> $OUT/bench_default.err < /dev/null
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
cd /tmp && export TMPDIR=/tmp
for W in config2 config3 genotype genotype_trio heuristic; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o t -- python $REPO/bench.py --workload $W --steps 3 --warmup 1 --configs off --pmc off --cpu-baseline-columns 0 > $OUT/trace_$W.log 2>&1 < /dev/null
  echo "trace $W rc=$?"
  find $OUT/trace_$W -name "*kernel_trace.csv" -size +2M -delete
  find $OUT/trace_$W -name "*_stats.csv" | head -3
done
du -sh $OUT

#!/bin/bash
# Round-4 evidence (run on the GPU box through gpurun from the repository root): the default bench line, the GPU test log,
# rocprofv3 --kernel-trace --stats of the main workloads (kernel-trace only: never combined with --pmc), the PMC CSVs bench.py kept.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r04
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json | head -c 200; echo
# the counter CSVs bench.py kept (one row per dispatch and counter: large) -- compressed, flat names; the scratch copies go (gpurun merges at most 64 MiB back)
for f in $(find gpurun_out/pmc_live -name "*_counter_collection.csv"); do
  name=$(echo ${f#gpurun_out/pmc_live/} | tr '/' '_')
  gzip -c $f > $OUT/pmc_$name.gz
done
rm -rf gpurun_out/pmc_live
cd /tmp && export TMPDIR=/tmp
for w in ${WHAMD_PROFILE_SET:-config2 blocks24 config1_x24 config3 config3_distrust heuristic_x32 genotype_trio}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o p -- python $REPO/bench.py --workload $w --sub --steps 3 --warmup 1 --pmc off --cpu-baseline-columns 0 --configs off > $OUT/trace_$w.log 2>&1
  f=$(find $OUT/trace_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/rocprof_kernel_stats_$w.csv && head -3 $f | cut -c1-200
  rm -rf $OUT/trace_$w
done
rm -f $OUT/trace_*.log
du -sh $OUT $REPO/gpurun_out

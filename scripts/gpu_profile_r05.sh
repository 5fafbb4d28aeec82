#!/bin/bash
# Round-5 evidence (run on the GPU box through gpurun from the repository root): the GPU test log, the default bench output (detail lines + the
# compact last line) and its detail file, rocprofv3 --kernel-trace --stats of the main workloads (kernel-trace only: never combined with --pmc),
# the PMC CSVs bench.py kept.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
T0=$SECONDS
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log; echo "pytest: $((SECONDS - T0)) s"
T0=$SECONDS
python bench.py --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json > $OUT/bench_default.out 2> $OUT/bench_default.err; echo "bench: $((SECONDS - T0)) s"
tail -1 $OUT/bench_default.out > $OUT/bench_default_last_line.json; wc -c $OUT/bench_default_last_line.json; head -c 400 $OUT/bench_default_last_line.json; echo
for f in $(find gpurun_out/pmc_live -name "*_counter_collection.csv"); do
  name=$(echo ${f#gpurun_out/pmc_live/} | tr '/' '_')
  gzip -c $f > $OUT/pmc_$name.gz
done
rm -rf gpurun_out/pmc_live
cd /tmp && export TMPDIR=/tmp
for w in ${WHAMD_PROFILE_SET:-config2 config1 blocks24 config1_x96 config3_x8 irregular config3 quartet_distrust}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o p -- python $REPO/bench.py --workload $w --sub --steps 3 --warmup 1 --pmc off --cpu-baseline-columns 0 --configs off > $OUT/trace_$w.log 2>&1
  f=$(find $OUT/trace_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/rocprof_kernel_stats_$w.csv && head -3 $f | cut -c1-200
  rm -rf $OUT/trace_$w
done
rm -f $OUT/trace_*.log
du -sh $OUT $REPO/gpurun_out

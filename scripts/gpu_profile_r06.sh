#!/bin/bash
# Round-6 evidence (run on the GPU box through gpurun from the repository root): the GPU test log, the default bench output (detail lines + the compact
# last line) and its detail file, the PMC CSVs bench.py kept, rocprofv3 --kernel-trace --stats of the main workloads (kernel-trace only: never combined with
# --pmc), the round's probes (create phases, wide tables, persistent barrier, host -> device copy rate), `bench.py --gpus 2 --oversubscribe`.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06
mkdir -p $OUT
T0=$SECONDS
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log; echo "pytest: $((SECONDS - T0)) s"
T0=$SECONDS
python bench.py --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json > $OUT/bench_default.out 2> $OUT/bench_default.err; echo "bench: $((SECONDS - T0)) s"
tail -1 $OUT/bench_default.out > $OUT/bench_default_last_line.json; wc -c $OUT/bench_default_last_line.json; head -c 600 $OUT/bench_default_last_line.json; echo
for f in $(find gpurun_out/pmc_live -name "*_counter_collection.csv"); do
  name=$(echo ${f#gpurun_out/pmc_live/} | tr '/' '_')
  gzip -c $f > $OUT/pmc_$name.gz
done
rm -rf gpurun_out/pmc_live
python bench.py --gpus 2 --oversubscribe --steps 3 --warmup 1 --detail-file $OUT/bench_two_ranks_detail.json > $OUT/bench_two_ranks_one_device.out 2> $OUT/bench_two_ranks_one_device.err; grep "bench rank" $OUT/bench_two_ranks_one_device.err | cut -c1-120
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 200000 20 > $OUT/create_phases_config2.txt 2>&1
WHAMD_DEBUG_TIMING=1 WHAMD_PLAN_THREADS=2 python scripts/gpu_create_timing.py 50000 15 > $OUT/create_phases_config1_2threads.txt 2>&1
python scripts/gpu_concurrent_create.py 96 16 2 > $OUT/concurrent_creates_96.txt 2>&1
python scripts/gpu_wide_ab.py 4000 > $OUT/wide_tables_ab.txt 2>&1
g++ -O2 -std=c++17 -pthread -o /tmp/r6q scripts/micro/r6_cpu_quota_probe.cpp && { /tmp/r6q; cat /sys/fs/cgroup/cpu.max; } > $OUT/cpu_quota_probe.txt 2>&1
g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl && R6_REPS=4 WHAMD_PLAN_THREADS=1 /tmp/r6ps whatshap_amd/libwhatshap_amd.so > $OUT/host_plan_scaling.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/r6pb scripts/micro/r6_persistent_barrier.hip && timeout 120 /tmp/r6pb > $OUT/persistent_barrier_probe.txt 2>&1
python scripts/micro/r6_h2d_rate.py > $OUT/h2d_rate.txt 2>&1
python scripts/gpu_shim_e2e.py 200000 20 > $OUT/shim_config2_pieces.txt 2>&1
# the round's late probes: what bounds concurrent creates (the same rate with the staging image not sent), creates under a running solve (upload streams against the
# tables' own streams), the pieces of a resident 96-table step, the time line of a fresh 96-table step
{ echo "== product library"; python scripts/gpu_create_rate_ab.py; echo "== staging image built, not sent (debug library, WHAMD_SKIP_SLAB_COPY=1: results invalid)"; WHAMD_USE_DEBUG_LIB=1 WHAMD_SKIP_SLAB_COPY=1 python scripts/gpu_create_rate_ab.py; echo "== product library, process not bound to one socket"; WHAMD_NO_BIND=1 WHAMD_RATE_SHAPES=32x1,64x1,96x1 python scripts/gpu_create_rate_ab.py; } > $OUT/create_rate_ab.txt 2>&1
{ echo "== upload streams (product)"; python scripts/gpu_create_under_solve.py; echo "== the tables' own streams (debug library, WHAMD_UPLOAD_ON_TABLE_STREAM=1)"; WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_ON_TABLE_STREAM=1 python scripts/gpu_create_under_solve.py; } > $OUT/create_under_solve.txt 2>&1
{ echo "== product (upload streams of the default priority)"; python scripts/gpu_wide_tables_concurrent.py; echo "== debug library, upload streams of the highest priority (the first version)"; WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_STREAMS_HIGH=1 python scripts/gpu_wide_tables_concurrent.py; } > $OUT/wide_tables_concurrent.txt 2>&1
WHAMD_DEBUG_TIMING=1 python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep\|wait_many of" > $OUT/group_step_pieces_96.txt
WHAMD_E2E_WINDOWS=96,48,32 python scripts/gpu_e2e_trace.py 96 50000 15 > $OUT/fresh_step_trace_96.txt 2>&1
python scripts/gpu_close_timing.py 50000 15 96 2>/dev/null > $OUT/release_and_close_96.txt
cd /tmp && export TMPDIR=/tmp
for w in ${WHAMD_PROFILE_SET:-config2 config1 blocks24 config1_x96 config3_x8 irregular irregular_x24 config_cov23 config3}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o p -- python $REPO/bench.py --workload $w --sub --steps 3 --warmup 1 --pmc off --cpu-baseline-columns 0 --configs off > $OUT/trace_$w.log 2>&1
  f=$(find $OUT/trace_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/rocprof_kernel_stats_$w.csv && head -3 $f | cut -c1-200
  rm -rf $OUT/trace_$w
done
rm -f $OUT/trace_*.log
du -sh $OUT $REPO/gpurun_out

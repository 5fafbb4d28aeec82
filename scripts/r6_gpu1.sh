set -x
cd /root/repo
mkdir -p gpurun_out/r6a
nproc; lscpu | head -20; cat /sys/kernel/mm/transparent_hugepage/enabled
g++ -O2 -pthread -o /tmp/mb scripts/micro/r6_pagefault_fill.cpp 2>/dev/null; for a in "1 0" "8 0" "1 1" "8 1"; do /tmp/mb $a | tail -1; done
python scripts/gpu_shim_e2e.py 200000 20 2>&1 | tee gpurun_out/r6a/shim_config2.txt
python scripts/gpu_shim_e2e.py 100000 15 trio 2>&1 | tee gpurun_out/r6a/shim_trio.txt
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 200000 20 2>&1 | tee gpurun_out/r6a/create_config2.txt
WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 50000 15 2>&1 | tee gpurun_out/r6a/create_config1.txt
WHAMD_DEBUG_TIMING=1 WHAMD_PLAN_THREADS=2 python scripts/gpu_create_timing.py 50000 15 2>&1 | tee gpurun_out/r6a/create_config1_2threads.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_heuristic.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6a/pytest_subset.txt

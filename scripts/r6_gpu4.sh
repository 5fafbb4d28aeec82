cd /root/repo
mkdir -p gpurun_out/r6d
python scripts/gpu_close_timing.py 50000 15 1 2>&1 | grep -v "^\[whamd timing\]   \|slot plan\|upload:" | tail -12
python scripts/gpu_close_timing.py 50000 15 24 2>&1 | grep -v "^\[whamd timing\]   \|slot plan\|upload:\|create:" | tail -40
echo ---- trio creates in a 16-CPU slice
taskset -c 0-7,128-135 env WHAMD_DEBUG_TIMING=1 python scripts/gpu_close_timing.py 100000 15 2 trio 2>&1 | grep -v "^\[whamd timing\]   " | tail -30

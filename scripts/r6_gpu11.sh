cd /root/repo
hipcc --offload-arch=gfx950 -O3 -o /tmp/r6pb scripts/micro/r6_persistent_barrier.hip && timeout 120 /tmp/r6pb
echo ---- trio create-rate child with timing
WHAMD_DEBUG_TIMING=1 timeout 300 python bench.py --create-rate-worker 0/8 --coverage 15 --variants 100000 --blocks 4 --path auto --trio 2>&1 | grep -v "^\[whamd timing\]   " | tail -30

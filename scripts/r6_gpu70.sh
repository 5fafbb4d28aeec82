cd /root/repo
WHAMD_DEBUG_TIMING=1 WHAMD_PLAN_THREADS=1 python scripts/gpu_create_timing.py 50000 15 2>&1 | tail -28

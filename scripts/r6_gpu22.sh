cd /root/repo
for th in 16 32 48 64 96; do echo "== WHAMD_PLAN_THREADS=$th"; WHAMD_PLAN_THREADS=$th WHAMD_DEBUG_TIMING=1 python scripts/gpu_create_timing.py 200000 20 2>&1 | grep -E "^create |create: flatten|entries, indexing|column ranges" | tail -4; done

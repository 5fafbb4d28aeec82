cd /root/repo
echo "== debug library, dedicated upload streams of NORMAL priority"
WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_STREAMS_NORMAL=1 python scripts/gpu_wide_tables_concurrent.py | tail -3
WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_STREAMS_NORMAL=1 python scripts/gpu_create_under_solve.py | tail -3
echo "== debug library, high priority (as the product)"
WHAMD_USE_DEBUG_LIB=1 python scripts/gpu_wide_tables_concurrent.py | tail -3
WHAMD_USE_DEBUG_LIB=1 python scripts/gpu_create_under_solve.py | tail -3

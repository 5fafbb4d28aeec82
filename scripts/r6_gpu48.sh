cd /root/repo
echo "product:"; python scripts/gpu_create_rate_ab.py 2>&1 | tail -6
echo "image built, not sent (debug library, results invalid):"; WHAMD_USE_DEBUG_LIB=1 WHAMD_SKIP_SLAB_COPY=1 python scripts/gpu_create_rate_ab.py 2>&1 | tail -6

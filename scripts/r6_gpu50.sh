cd /root/repo
WHAMD_DEBUG_TIMING=1 python scripts/gpu_concurrent_create.py --inner 96 16 2 2> gpurun_out/cc50.err | tail -3
grep "upload: plan" gpurun_out/cc50.err | tail -96 | awk '{print $11, $12, $13, $14, $15, $16}' | sort | uniq -c | sort -rn | head -12
grep "upload: plan" gpurun_out/cc50.err | tail -5

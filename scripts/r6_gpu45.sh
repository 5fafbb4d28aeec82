cd /root/repo
WHAMD_E2E_WINDOWS=96,48,32,24 python scripts/gpu_e2e_trace.py 96 50000 15 > gpurun_out/e2e45.out 2>&1
grep "^rep [12]" gpurun_out/e2e45.out || tail -20 gpurun_out/e2e45.out

cd /root/repo
mkdir -p gpurun_out/r6soak
timeout 1500 python scripts/gpu_group_soak.py > gpurun_out/r6soak/group_soak.txt 2>&1; tail -4 gpurun_out/r6soak/group_soak.txt
timeout 900 python scripts/gpu_irregular_soak.py > gpurun_out/r6soak/irregular_soak.txt 2>&1; tail -3 gpurun_out/r6soak/irregular_soak.txt
WHAMD_SOAK_BLOCKS=40 timeout 1500 python scripts/gpu_soak.py > gpurun_out/r6soak/soak.txt 2>&1; tail -4 gpurun_out/r6soak/soak.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "headline or group or parity" 2>&1 | tail -2

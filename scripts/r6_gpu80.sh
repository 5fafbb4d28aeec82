cd /root/repo
mkdir -p gpurun_out/r6soak5
timeout 1500 python scripts/gpu_group_soak.py > gpurun_out/r6soak5/group_soak.txt 2>&1; tail -2 gpurun_out/r6soak5/group_soak.txt
timeout 900 python scripts/gpu_irregular_soak.py > gpurun_out/r6soak5/irregular_soak.txt 2>&1; tail -1 gpurun_out/r6soak5/irregular_soak.txt
WHAMD_SOAK_BLOCKS=100 timeout 1500 python scripts/gpu_soak.py > gpurun_out/r6soak5/soak.txt 2>&1; grep -i mismatch gpurun_out/r6soak5/soak.txt | tail -4
timeout 900 python scripts/gpu_pedslot_check.py > gpurun_out/r6soak5/pedslot_check.txt 2>&1; tail -2 gpurun_out/r6soak5/pedslot_check.txt

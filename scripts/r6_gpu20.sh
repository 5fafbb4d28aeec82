cd /root/repo
python scripts/gpu_create_timing_ped.py 50000 13 quartet 2>&1 | tail -24
python scripts/gpu_create_timing_ped.py 100000 15 trio 2>&1 | tail -24 | grep -v "^\[whamd timing\]   upload: slot\|X runs"

cd /root/repo
python scripts/gpu_close_timing.py 50000 15 96 2>/dev/null | tail -3

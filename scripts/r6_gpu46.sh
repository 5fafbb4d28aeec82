cd /root/repo
python scripts/gpu_create_under_solve.py 2>&1 | tail -5

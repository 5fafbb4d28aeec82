cd /root/repo
python scripts/gpu_concurrent_create.py 96 16 2
python scripts/gpu_concurrent_create.py 96 1 2
python scripts/gpu_concurrent_create.py 96 64 1

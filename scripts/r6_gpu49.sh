cd /root/repo
echo "== 32 workers x 1 thread"; python scripts/gpu_concurrent_create.py 96 32 1 2>&1 | tail -26
echo "== 16 workers x 2 thread"; python scripts/gpu_concurrent_create.py 96 16 2 2>&1 | tail -26

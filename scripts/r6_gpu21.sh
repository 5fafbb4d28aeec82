cd /root/repo
echo "== one process"; python scripts/gpu_concurrent_create.py 96 16 2 | head -3
echo "== two processes at once (each 96 creates on 16 x 2 threads)"
python scripts/gpu_concurrent_create.py 96 16 2 > /tmp/a.txt & python scripts/gpu_concurrent_create.py 96 16 2 > /tmp/b.txt; wait; head -3 /tmp/a.txt; head -3 /tmp/b.txt
echo "== four processes at once (each 48 creates on 8 x 2 threads)"
for i in 1 2 3 4; do python scripts/gpu_concurrent_create.py 48 8 2 > /tmp/c$i.txt & done; wait; for i in 1 2 3 4; do sed -n 2,3p /tmp/c$i.txt; done

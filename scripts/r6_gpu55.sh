cd /root/repo
echo unbound; WHAMD_NO_BIND=1 python scripts/gpu_create_rate_ab.py 2>&1 | tail -6
lscpu | grep -i "numa\|socket\|model name" | head -8

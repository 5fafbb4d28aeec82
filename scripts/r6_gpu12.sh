cd /root/repo
WHAMD_DEBUG_TIMING=1 timeout 300 python bench.py --create-rate-worker 0/8 --coverage 15 --variants 100000 --blocks 8 --path auto --trio 2>&1 | grep -v "^\[whamd timing\]   " | grep -v "slot plan" | tail -42

cd /root/repo
echo "== pooled host memory"; python scripts/host_plan_scaling.py
echo "== WHAMD_HOST_POOL_MB=0"; WHAMD_HOST_POOL_MB=0 python scripts/host_plan_scaling.py

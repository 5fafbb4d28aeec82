cd /root/repo
g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl
echo "== bound to node 0 (taskset 0-63,128-191), pooled host memory"; WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so
echo "== the same, WHAMD_HOST_POOL_MB=0"; WHAMD_HOST_POOL_MB=0 WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so

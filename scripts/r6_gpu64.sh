cd /root/repo
g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl
for w in 32 64; do WHAMD_POOL_STATS=1 R6_REPS=5 R6_ONLY=$w WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so 2>&1 | tail -4; done

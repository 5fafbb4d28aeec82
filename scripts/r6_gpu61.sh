cd /root/repo
g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl
run() { echo "== $1"; shift; for w in 32 64; do env "$@" R6_ONLY=$w WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so; done; }
run "default" X=1
run "no trim, top pad 64 MB" MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=67108864
run "no trim, top pad 64 MB, mmap threshold 1 GB (pool off)" MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=67108864 MALLOC_MMAP_THRESHOLD_=1073741824 WHAMD_HOST_POOL_MB=0
run "arena_max 256" MALLOC_ARENA_MAX=256
run "no huge page advice" WHAMD_NO_HUGEPAGES=1
grep -i "thp\|AnonHuge" /proc/meminfo | head -3; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag

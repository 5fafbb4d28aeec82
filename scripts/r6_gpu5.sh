cd /root/repo
for envs in "" "MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=8589934592 MALLOC_TOP_PAD_=268435456" "MALLOC_ARENA_MAX=1 MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=8589934592"; do
for w in config1_x24 config1_x96; do env $envs python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>&1 >/dev/null | grep "bench rank" | sed "s/blocks \[[^]]*\]//" | sed "s/^/[$envs] $w: /"; done; done

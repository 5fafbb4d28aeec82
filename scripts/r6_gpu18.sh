cd /root/repo
for e in "" "WHAMD_UPLOAD_STREAM=1"; do
echo "== $e"
env $e python scripts/gpu_concurrent_create.py 96 16 2 | head -3
env $e python scripts/gpu_concurrent_create.py 96 32 1 | head -3
env $e python bench.py --workload config1_x96 --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>&1 >/dev/null | grep "bench rank" | sed "s/blocks \[[^]]*\]//"
env $e python bench.py --configs off --pmc off --cpu-baseline-columns 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "bench rank"
done

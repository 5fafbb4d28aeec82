cd /root/repo
hipcc --offload-arch=gfx950 -O2 -o /tmp/meminfo scripts/micro/r6_meminfo_cost.hip -pthread && /tmp/meminfo
python scripts/gpu_close_timing.py 50000 15 96 2>/dev/null | tail -2
for w in config1_x96 config1_x24 blocks24; do
python bench.py --workload $w --configs off --pmc off --cpu-baseline-columns 0 2> gpurun_out/bench54_$w.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value']), round(d['ms_per_step'],1), json.dumps(d.get('value_resident'))[:60], json.dumps(d.get('per_rank'))[:400])"
done

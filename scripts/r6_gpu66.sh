cd /root/repo
for w in config2 config1_x96 config1_x24 blocks24 irregular_x24 config3_x8; do
python bench.py --workload $w --configs off --pmc off --cpu-baseline-columns 0 2> gpurun_out/bench66_$w.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value']), round(d['ms_per_step'],1), json.dumps(d.get('value_resident'))[:60], json.dumps(d.get('per_rank'))[:460])"
done
cat /sys/fs/cgroup/cpu.stat | grep -i thrott

cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in config1_x96 config1_x24 blocks24; do
python bench.py --workload $w --configs off --pmc off --cpu-baseline-columns 0 2> gpurun_out/bench51_$w.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value']), round(d['ms_per_step'],1), json.dumps(d.get('value_resident'))[:120], json.dumps(d.get('per_rank'))[:700])"
done

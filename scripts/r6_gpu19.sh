cd /root/repo
mkdir -p gpurun_out/r6l
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in config3_distrust quartet_distrust quartet config3; do python bench.py --workload $w --sub --pmc off --steps 5 --warmup 2 2>gpurun_out/r6l/$w.err | tail -1 > gpurun_out/r6l/$w.json; grep "bench rank" gpurun_out/r6l/$w.err | sed "s/blocks \[[^]]*\]//"; python -c "
import json,sys; d=json.load(open('gpurun_out/r6l/$w.json')); print('$w', 'value', round(d['value']), 'resident', round(d['value_resident']['value']), 'ident', d.get('identical_to_reference'))"; done
python scripts/gpu_pedslot_check.py 2>&1 | tail -5

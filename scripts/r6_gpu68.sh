cd /root/repo
python bench.py --workload config1_x96 --configs off --pmc off --cpu-baseline-columns 0 --detail-file gpurun_out/d68.json > gpurun_out/b68.out 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/d68.json'))
print(round(d['value']), d['ms_per_step'])
for r in d['host_shapes_tried']: print(r)
print(d['per_rank'])
PY

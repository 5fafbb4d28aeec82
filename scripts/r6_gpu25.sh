cd /root/repo
for r in 1 0; do echo "== WHAMD_BENCH_RELEASE=$r"; for w in config1_x96 config1_x24; do WHAMD_BENCH_RELEASE=$r python bench.py --workload $w --sub --pmc off --cpu-baseline-columns 0 --steps 5 --warmup 2 2>&1 >/tmp/o.json | grep "bench rank" | sed "s/blocks \[[^]]*\]//"; python -c "
import json; d=json.load(open('/tmp/o.json')); print('$w value', round(d['value']), 'other_ms', d['per_rank'][0].get('other_ms'))"; done; done

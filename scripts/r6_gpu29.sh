cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/tr -o p -- python /root/repo/scripts/gpu_concurrent_create.py --inner 96 16 2 > /tmp/tr.log 2>&1
tail -3 /tmp/tr.log
find /tmp/tr -name "*stats*.csv" | head; for f in $(find /tmp/tr -name "*memory_copy_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-220; done
f=$(find /tmp/tr -name "*memory_copy_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
h2d = [r for r in rows if 'HOST_TO_DEVICE' in (r.get('Direction') or r.get('Name') or '').upper() or 'H2D' in (r.get('Direction') or '').upper()]
print(len(rows), 'copies', len(h2d), 'h2d; columns:', list(rows[0].keys()))
big = [r for r in rows if float(r.get('Bytes') or r.get('Size') or 0) > 4e6] if rows else []
import statistics
d = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3 for r in big]
b = [float(r.get('Bytes') or r.get('Size')) for r in big]
if d:
    print('big copies', len(d), 'median us', statistics.median(d), 'median MB', statistics.median(b) / 1e6, 'median GB/s', statistics.median([x / y / 1e3 for x, y in zip(b, d)]))
    t0 = min(float(r['Start_Timestamp']) for r in big); t1 = max(float(r['End_Timestamp']) for r in big)
    print('span of big copies ms', (t1 - t0) / 1e6, 'sum of durations ms', sum(d) / 1e3)
PY

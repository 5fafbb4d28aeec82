cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr2 && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr2 -o p -- python /root/repo/scripts/gpu_group_step_pieces.py 96 15 50000 > /tmp/tr2.log 2>&1
grep "^rep" /tmp/tr2.log | sed 's/; device.*finish per table/ finish per table/'
python3 - <<'PY'
import csv, glob
k = list(csv.DictReader(open(glob.glob('/tmp/tr2/**/*kernel_trace.csv', recursive=True)[0])))
c = list(csv.DictReader(open(glob.glob('/tmp/tr2/**/*memory_copy_trace.csv', recursive=True)[0])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1][:24]) for r in k] + [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy_' + r['Direction'][-16:]) for r in c]
ev.sort()
# steps: group kernels separated by > 5 ms of no slot_groupx
g = [e for e in ev if e[2].startswith('slot_groupx')]
steps, cur = [], [g[0]]
for e in g[1:]:
    if e[0] - cur[-1][1] > 5_000_000: steps.append(cur); cur = [e]
    else: cur.append(e)
steps.append(cur)
for si, st in enumerate(steps[-8:]):
    t0, t1 = st[0][0], st[-1][1]
    nxt = steps[-8:][si + 1][0][0] if si + 1 < len(steps[-8:]) else 10**30
    tail = [e for e in ev if e[0] >= t1 and e[0] < nxt and not e[2].startswith('slot_groupx')]
    by = {}
    for e in tail:
        d = by.setdefault(e[2], [0, 0, 10**30, 0]); d[0] += 1; d[1] += e[1] - e[0]; d[2] = min(d[2], e[0]); d[3] = max(d[3], e[1])
    print('step %d: forward span %.1f ms (%d launches); tail until %.1f ms after the last launch:' % (si, (t1 - t0) / 1e6, len(st), (max(e[1] for e in tail) - t1) / 1e6 if tail else 0))
    for name, d in sorted(by.items(), key=lambda kv: kv[1][2]):
        print('     %-26s x%4d busy %.2f ms, from +%.2f to +%.2f ms' % (name, d[0], d[1] / 1e6, (d[2] - t1) / 1e6, (d[3] - t1) / 1e6))
PY

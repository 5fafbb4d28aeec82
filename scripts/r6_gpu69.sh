cd /root/repo
g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl
R6_ONLY=1 WHAMD_DEBUG_TIMING=1 WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so 2> /tmp/err_1.txt | head -1
python - <<'PY'
import re, collections
sums = collections.defaultdict(float); cnt = collections.Counter()
for line in open("/tmp/err_1.txt"):
    m = re.match(r"\[whamd timing\]   flatten: (.+?) ([0-9.]+) ms", line.strip())
    if m: sums["flatten: " + m.group(1)] += float(m.group(2)); cnt["flatten: " + m.group(1)] += 1
    m = re.match(r"\[whamd timing\] slot plan: setup ([0-9.]+) ms, column ranges ([0-9.]+) ms, concatenation ([0-9.]+) ms, layouts ([0-9.]+) ms", line.strip())
    if m:
        for name, v in zip(("plan: setup", "plan: column ranges", "plan: concatenation", "plan: layouts"), m.groups()): sums[name] += float(v); cnt[name] += 1
for k in sums: print(f"  {k:70s} {sums[k] / cnt[k]:7.2f} ms")
PY

cd /root/repo
g++ -O2 -std=c++17 -pthread -I include -o /tmp/r6ps scripts/micro/r6_plan_scaling.cpp -ldl
vm() { grep -E "^(thp_fault_alloc|thp_fault_fallback|compact_stall|pgfault|thp_collapse_alloc) " /proc/vmstat | tr '\n' ' '; }
for i in 1 2 3 4 5; do
  a=$(vm); R6_ONLY=32 WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so; b=$(vm)
  python3 - "$a" "$b" <<'PY'
import sys
a=sys.argv[1].split(); b=sys.argv[2].split()
print("   default:", " ".join(f"{a[i]} +{int(b[i+1])-int(a[i+1])}" for i in range(0,len(a),2)))
PY
done
for i in 1 2 3 4 5; do
  a=$(vm); WHAMD_NO_HUGEPAGES=1 R6_ONLY=32 WHAMD_PLAN_THREADS=1 taskset -c 0-63,128-191 /tmp/r6ps whatshap_amd/libwhatshap_amd.so; b=$(vm)
  python3 - "$a" "$b" <<'PY'
import sys
a=sys.argv[1].split(); b=sys.argv[2].split()
print("   no advice:", " ".join(f"{a[i]} +{int(b[i+1])-int(a[i+1])}" for i in range(0,len(a),2)))
PY
done

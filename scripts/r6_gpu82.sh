cd /tmp && export TMPDIR=/tmp
for w in config3 quartet; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$w -o p -- python /root/repo/bench.py --workload $w --sub --steps 3 --warmup 1 --pmc off --cpu-baseline-columns 0 --configs off > /tmp/tr_$w.log 2>&1
f=$(find /tmp/tr_$w -name "*kernel_stats.csv" | head -1); grep "pedslot_tables\|Name" $f | cut -c1-60,150-260
done

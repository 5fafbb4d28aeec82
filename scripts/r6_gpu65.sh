cd /root/repo
g++ -O2 -std=c++17 -pthread -o /tmp/spin scripts/micro/r6_cpu_quota_probe.cpp && /tmp/spin
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8; cat /proc/self/cgroup | head -3; nproc

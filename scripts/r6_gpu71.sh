cd /root/repo
hipcc --offload-arch=gfx950 -O2 -o /tmp/pew scripts/micro/r6_priority_event_wait.hip && /tmp/pew

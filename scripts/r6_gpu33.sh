cd /root/repo
for e in "" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=2" ""; do echo "== $e"; env $e python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep" | tail -6 | sed 's/; device.*finish per table/ finish per table/' | awk '{print $2, $3, $4, $5, $6, $7, $8}' | tr '\n' ' '; echo; done

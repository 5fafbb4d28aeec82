cd /root/repo
for e in "" "WHAMD_WAIT_ONE_PHASE=1" "" "WHAMD_WAIT_ONE_PHASE=1"; do echo "== $e"; env WHAMD_USE_DEBUG_LIB=1 $e python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep" | tail -6 | sed 's/; device.*finish per table/ finish per table/'; done

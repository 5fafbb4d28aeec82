cd /root/repo
for e in "" "WHAMD_TAIL_OWN_STREAM=1" ""; do echo "== $e"; env WHAMD_USE_DEBUG_LIB=1 $e python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep" | tail -6 | sed 's/; device.*finish per table/ finish per table/' | awk '{print $2, $3, $4, $5, $6, $7, $8}' | tr '\n' ' '; echo; done
timeout 900 python -m pytest tests -m gpu -x -q -k "group or headline" 2>&1 | tail -2

cd /root/repo
WHAMD_DEBUG_TIMING=1 python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep\|wait_many of" | sed 's/device forward.*device first event/device first event/' | tail -8
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

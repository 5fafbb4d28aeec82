cd /root/repo
python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep" | sed 's/device forward.*device first event/device first event/' | tail -8

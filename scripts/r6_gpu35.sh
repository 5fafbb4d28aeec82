cd /root/repo
python scripts/gpu_group_step_pieces.py 96 15 50000 2>&1 | grep "^rep\|wait_many of" | sed 's/; device.*finish per table/ finish per table/' | tail -16

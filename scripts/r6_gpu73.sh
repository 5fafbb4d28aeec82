cd /root/repo
bash scripts/r6_gpu69.sh 2>&1 | grep "threads\|entries\|column ranges\|layouts"
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_parity.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -2

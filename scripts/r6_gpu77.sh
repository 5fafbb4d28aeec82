cd /root/repo
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_untrusted_group.py -m gpu -x -q 2>&1 | tail -2

cd /root/repo
echo "upload streams:"; python scripts/gpu_create_under_solve.py 2>&1 | tail -3
echo "table's own stream (debug library switch):"; WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_ON_TABLE_STREAM=1 python scripts/gpu_create_under_solve.py 2>&1 | tail -3
echo "upload streams, debug library:"; WHAMD_USE_DEBUG_LIB=1 python scripts/gpu_create_under_solve.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -2

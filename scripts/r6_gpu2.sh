set -x
cd /root/repo
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "bench_with_two_ranks or shim or switching" 2>&1 | tail -5 | tee gpurun_out/r6b/pytest_subset.txt
( time python bench.py ) > gpurun_out/r6b/bench_default.out 2> gpurun_out/r6b/bench_default.err
tail -c 7000 gpurun_out/r6b/bench_default.out
tail -20 gpurun_out/r6b/bench_default.err
cp gpurun_out/bench_detail.json gpurun_out/r6b/bench_detail.json

cd /root/repo
mkdir -p gpurun_out/r6h
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r6h/bench_default.out 2> gpurun_out/r6h/bench_default.err
tail -c 6500 gpurun_out/r6h/bench_default.out
tail -8 gpurun_out/r6h/bench_default.err
cp gpurun_out/bench_detail.json gpurun_out/r6h/bench_detail.json

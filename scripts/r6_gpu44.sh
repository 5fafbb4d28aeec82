cd /root/repo
python bench.py > gpurun_out/bench44.out 2> gpurun_out/bench44.err
tail -1 gpurun_out/bench44.out | cut -c1-3000

cd /tmp && export TMPDIR=/tmp
cd /root/repo
WHAMD_RATE_SHAPES=32x1 rocprofv3 --hip-trace --stats -d gpurun_out/hiptrace56 -o t -- python scripts/gpu_create_rate_ab.py > gpurun_out/hiptrace56.out 2>&1
tail -2 gpurun_out/hiptrace56.out
f=$(find gpurun_out/hiptrace56 -name "*hip_api_stats.csv" | head -1); echo $f; head -25 $f
find gpurun_out/hiptrace56 -name "*hip_api_trace.csv" -size +60M -delete

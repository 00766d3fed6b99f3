mkdir -p gpurun_out/j9
python tools/profiling/host_profile.py 1 > gpurun_out/j9/host_1sweep.txt 2>&1
python tools/profiling/host_profile.py 10 > gpurun_out/j9/host_10sweep.txt 2>&1
head -75 gpurun_out/j9/host_1sweep.txt | cut -c1-180

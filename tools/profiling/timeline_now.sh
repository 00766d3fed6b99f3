export TMPDIR=/tmp
rm -rf gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block > /dev/null 2>&1
mkdir -p gpurun_out/r5
python tools/profiling/frame_timeline.py gpurun_out/prof_tmp/fsf_results.db 7 > gpurun_out/r5/frame_timeline_now.txt
head -3 gpurun_out/r5/frame_timeline_now.txt
rm -rf gpurun_out/prof_tmp

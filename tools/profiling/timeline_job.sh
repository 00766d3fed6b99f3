export TMPDIR=/tmp
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tl -o fsf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like > /dev/null 2>&1
python tools/profiling/frame_timeline.py gpurun_out/prof_tl/fsf_results.db 7 > gpurun_out/frame_timeline.txt
rm -rf gpurun_out/prof_tl
python tools/profiling/launch_sites.py > gpurun_out/launch_sites.txt 2>&1
tail -5 gpurun_out/frame_timeline.txt

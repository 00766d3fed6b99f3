timeout 600 python -m pytest tests -m gpu -x -q -k 'voxel2point or neck or vote' 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
rm -rf gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --no-h2d --serial > /dev/null 2>&1
python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db 7 x | grep -E "vote_centers|voxel2point|total kernel"
rm -rf gpurun_out/prof_tmp

export TMPDIR=/tmp
rm -rf gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py --dataset av2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --serial > /dev/null 2>&1
python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db 7 x | head -45 | cut -c1-150
rm -rf gpurun_out/prof_tmp
for i in 1 2 3; do python bench.py --dataset av2 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block | grep -o "\"ms_per_step\": [0-9.]*"; done
python bench.py --dataset av2 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --serial | grep -o "\"ms_per_step\": [0-9.]*"

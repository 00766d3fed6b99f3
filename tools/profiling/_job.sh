mkdir -p gpurun_out/j4
python tools/profiling/train_sites.py > gpurun_out/j4/train_sites.txt 2>&1
python tools/profiling/sir_bench.py > gpurun_out/j4/sir_bench.txt 2>&1
cat gpurun_out/j4/train_sites.txt | cut -c1-230

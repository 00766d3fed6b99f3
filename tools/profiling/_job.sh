mkdir -p gpurun_out/j5
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "rulebook or point_pool or pool" 2>&1 | tail -5 > gpurun_out/j5/tests_a.txt
timeout 1800 python -m pytest tests/test_plugin_gpu.py tests/test_e2e_agreement_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/j5/tests_b.txt
bash tools/profiling/ab_bench.sh "FSF_REFINE_DIRECT=0" "FSF_REFINE_DIRECT=1" > gpurun_out/j5/ab.txt 2>&1
python tools/profiling/sync_sites.py > gpurun_out/j5/sync.txt 2>&1
python tools/profiling/stage_times.py > gpurun_out/j5/stages.txt 2>&1
cat gpurun_out/j5/tests_a.txt gpurun_out/j5/tests_b.txt gpurun_out/j5/ab.txt gpurun_out/j5/sync.txt gpurun_out/j5/stages.txt

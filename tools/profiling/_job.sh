mkdir -p gpurun_out/j15
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "group_pairs" 2>&1 | tail -4 > gpurun_out/j15/tests_a.txt
timeout 2400 python -m pytest tests/test_plugin_gpu.py tests/test_e2e_agreement_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/j15/tests_b.txt
bash tools/profiling/ab_bench.sh "FSF_FUSION_ADD_FUSED=0" "FSF_FUSION_ADD_FUSED=1" > gpurun_out/j15/ab.txt 2>&1
python tools/profiling/stage_times.py > gpurun_out/j15/stages.txt 2>&1
cat gpurun_out/j15/tests_a.txt gpurun_out/j15/tests_b.txt gpurun_out/j15/ab.txt; grep -v amdgpu gpurun_out/j15/stages.txt

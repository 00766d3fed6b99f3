export TMPDIR=/tmp
python tools/profiling/seg_calls.py 2>/dev/null > gpurun_out/seg_calls.txt
rm -rf gpurun_out/prof_seg
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_seg -o fsf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like > /dev/null 2>&1
python tools/profiling/dispatch_list.py gpurun_out/prof_seg/fsf_results.db 7 seg_reduce_kernel seg_fixup_long sir_input_kernel > gpurun_out/seg_dispatches.txt
rm -rf gpurun_out/prof_seg
cat gpurun_out/seg_calls.txt; cat gpurun_out/seg_dispatches.txt

mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py --train --no-roofline --steps 5 --warmup 2 > gpurun_out/r5_trainprof_bench.json 2>gpurun_out/r5_trainprof.err
python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db 7 "train" > gpurun_out/r5_kernel_stats_train_step_pre.txt 2>&1
rm -rf gpurun_out/prof_tmp
python tools/profiling/train_ops.py 70 > gpurun_out/r5_train_step_ops_pre.txt 2>&1
head -45 gpurun_out/r5_kernel_stats_train_step_pre.txt | cut -c1-170

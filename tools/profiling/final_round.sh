#!/bin/bash
# usage (on the GPU box): tools/profiling/final_round.sh <tag>   -> gpurun_out/final_<tag>/*  (the files copied into profiles/)
tag=$1
out=gpurun_out/final_$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --train --steps 10 --warmup 3 > $out/bench_train.json 2>> $out/bench.err
python bench.py --train --frames-per-gpu 2 --steps 6 --warmup 2 > $out/bench_train_bs2.json 2>> $out/bench.err
python bench.py --dataset av2 --no-cpu-baseline > $out/bench_av2.json 2>> $out/bench.err
bash tools/profiling/run_prof.sh fwd_$tag > /dev/null 2>&1
cp gpurun_out/fwd_${tag}_kernels.txt $out/kernel_stats_full_forward.txt
cp gpurun_out/fwd_${tag}_bench.json $out/bench_under_rocprof.json
rm -rf gpurun_out/prof_tr
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tr -o fsf -- python bench.py --train --steps 5 --warmup 2 > $out/bench_train_under_rocprof.json 2>> $out/bench.err
python tools/profiling/prof_summary.py gpurun_out/prof_tr/fsf_results.db 7 "rocprofv3 --kernel-trace --stats -- python bench.py --train --steps 5 --warmup 2 ($tag)" > $out/kernel_stats_train_step.txt
rm -rf gpurun_out/prof_tr
bash tools/profiling/pmc_traffic.sh $tag > /dev/null 2>&1
cp gpurun_out/${tag}_pmc_traffic.json $out/pmc_traffic.json
tail -c 600 $out/bench.json; echo; tail -c 300 $out/bench_train.json; echo; tail -c 300 $out/bench_train_bs2.json; echo; tail -c 300 $out/bench_av2.json

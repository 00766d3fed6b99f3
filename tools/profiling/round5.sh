#!/bin/bash
# usage (on the GPU box): tools/profiling/round5.sh <commit>   -> gpurun_out/r5/*  (the files copied into profiles/ as r5_*)
# Every text file starts with the commit it was taken at; the JSON lines carry it as "commit" (bench.py reads FSF_COMMIT).
commit=$1
out=gpurun_out/r5
mkdir -p $out
export TMPDIR=/tmp FSF_COMMIT=$commit
hdr() { echo "# commit $commit, one MI355X box ($(hostname)), $(date -u +%Y-%m-%dT%H:%MZ)"; }
prof() {  # <name> <frames> <bench args...>: rocprofv3 kernel trace of a bench run -> kernel table (+ timeline of the last frame)
  name=$1; frames=$2; shift; shift
  rm -rf gpurun_out/prof_tmp
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py "$@" > $out/bench_under_rocprof_$name.json 2>> $out/bench.err
  { hdr; python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db $frames "rocprofv3 --kernel-trace --stats -- python bench.py $*"; } > $out/kernel_stats_$name.txt
  if [ "$name" != "train_step" ]; then { hdr; python tools/profiling/frame_timeline.py gpurun_out/prof_tmp/fsf_results.db $frames; } > $out/frame_timeline_$name.txt; fi
  rm -rf gpurun_out/prof_tmp
}
FWD="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block"
prof full_forward 7 $FWD
prof full_forward_serial 7 $FWD --serial
prof config2_1sweep 7 $FWD --sweeps 1
prof train_step 7 --train --no-roofline --steps 5 --warmup 2
# the bench lines read their in-situ table / traced launch count from profiles/: the box's copy gets the files just taken (the same
# files are committed as profiles/r5_* afterwards)
cp $out/kernel_stats_full_forward_serial.txt profiles/r5_kernel_stats_full_forward_serial.txt
cp $out/kernel_stats_full_forward.txt profiles/r5_kernel_stats_full_forward.txt
bash tools/profiling/pmc_traffic.sh r5 > /dev/null 2>&1
cp gpurun_out/r5_pmc_traffic.json $out/pmc_traffic.json
cp gpurun_out/r5_pmc_traffic.json profiles/r5_pmc_traffic.json
python bench.py > $out/bench_final.json 2>> $out/bench.err
python bench.py --sweeps 1 --no-cpu-baseline --no-train-block > $out/bench_config2_1sweep.json 2>> $out/bench.err
python bench.py --train --steps 10 --warmup 3 > $out/bench_train.json 2>> $out/bench.err
python bench.py --train --hot-path-only --steps 10 --warmup 3 --no-roofline > $out/bench_train_query_stages_only.json 2>> $out/bench.err
python bench.py --train --frames-per-gpu 2 --steps 6 --warmup 2 > $out/bench_train_bs2.json 2>> $out/bench.err
python bench.py --train --dataset av2 --steps 10 --warmup 3 > $out/bench_train_av2.json 2>> $out/bench.err
python bench.py --dataset av2 --no-cpu-baseline > $out/bench_av2.json 2>> $out/bench.err
python bench.py --trained-like --no-cpu-baseline > $out/bench_trained_like.json 2>> $out/bench.err
{ hdr; python tools/profiling/planes_layers.py 2>/dev/null; } > $out/spconv_layers_k9b_vs_k9d.txt
{ hdr; python tools/profiling/k22_calls.py 2>/dev/null; } > $out/k22_family_calls.txt
{ hdr; python tools/profiling/sync_sites.py 2>/dev/null; } > $out/host_sync_sites.txt
{ hdr; python tools/profiling/aten_sites.py 90 2>/dev/null; } > $out/aten_sites.txt
{ hdr; python tools/profiling/stage_times.py 2>/dev/null; } > $out/stage_times.txt
{ hdr; python tools/profiling/train_ops.py 60 2>/dev/null; } > $out/train_step_ops.txt
{ hdr; bash tools/profiling/ab_bench.sh "FSF_K22H=0 FSF_SCS_XCD=0" "FSF_K22H=1" 2>/dev/null; } > $out/ab_k22h_k9b_xcd.txt
{ hdr; bash tools/profiling/ab_env_kernel_stats.sh FSF_K22F "linear_norm_act_kernel" 2>/dev/null; } > $out/ab_k22f_kernel_stats.txt
{ hdr; python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n"; } > $out/sir_k21_k22_microbench.txt
tail -c 600 $out/bench_final.json; echo; tail -c 300 $out/bench_train.json; echo; head -5 $out/kernel_stats_full_forward.txt

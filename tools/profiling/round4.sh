#!/bin/bash
# usage (on the GPU box): tools/profiling/round4.sh <commit>   -> gpurun_out/r4/*  (the files copied into profiles/ as r4_*)
# Every text file starts with the commit it was taken at; the JSON lines carry it as "commit" (bench.py reads FSF_COMMIT).
commit=$1
out=gpurun_out/r4
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
# files are committed as profiles/r4_* afterwards)
cp $out/kernel_stats_full_forward_serial.txt profiles/r4_kernel_stats_full_forward_serial.txt
cp $out/kernel_stats_full_forward.txt profiles/r4_kernel_stats_full_forward.txt
python bench.py > $out/bench_final.json 2>> $out/bench.err
python bench.py --sweeps 1 --no-cpu-baseline --no-train-block > $out/bench_config2_1sweep.json 2>> $out/bench.err
python bench.py --train --steps 10 --warmup 3 > $out/bench_train.json 2>> $out/bench.err
python bench.py --train --frames-per-gpu 2 --steps 6 --warmup 2 > $out/bench_train_bs2.json 2>> $out/bench.err
python bench.py --dataset av2 --no-cpu-baseline > $out/bench_av2.json 2>> $out/bench.err
python bench.py --trained-like --no-cpu-baseline > $out/bench_trained_like.json 2>> $out/bench.err
bash tools/profiling/pmc_traffic.sh r4 > /dev/null 2>&1
cp gpurun_out/r4_pmc_traffic.json $out/pmc_traffic.json
{ hdr; echo "# K21 / K22 / K22s / segmented max at the LiDAR-query SIR stack's shapes: round-3 end state (ab_prev) vs this commit, same box, twice"
  for i in 1 2; do echo "[round-3 end state]"; FSF_ROOT=ab_prev python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n"; echo "[this commit]"; python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n"; done; } > $out/sir_kernels_r3_vs_r4.txt
{ hdr; for v in 0 1; do echo "[FSF_BWD_SPLIT=$v] (0 = K10 on the fp32 matrix pipe, 1 = K10p on bf16 x 6)"; FSF_BWD_SPLIT=$v python tools/profiling/bwd_weight_layers.py 2>/dev/null; done; } > $out/spconv_bwd_weight_k10_vs_k10p.txt
{ hdr; python tools/profiling/planes_layers.py 2>/dev/null; } > $out/spconv_layers_k9b_vs_k9d.txt
{ hdr; python tools/profiling/inflight_probe.py 24 2>/dev/null; } > $out/frames_in_flight_probe.txt
{ hdr; python tools/profiling/sync_sites.py 2>/dev/null; } > $out/host_sync_sites.txt
{ hdr; python tools/profiling/aten_sites.py 90 2>/dev/null; } > $out/aten_sites.txt
{ hdr; python tools/profiling/stage_times.py 2>/dev/null; } > $out/stage_times.txt
{ hdr; python tools/profiling/train_ops.py 60 2>/dev/null; } > $out/train_step_ops.txt
{ hdr; bash tools/profiling/ab_bench.sh "FSF_SIR_SORTED=0" "FSF_SIR_SORTED=1" 2>/dev/null; } > $out/ab_sir_sorted.txt
{ hdr; for v in 0 1 0 1; do FSF_BWD_SPLIT=$v python bench.py --train --no-roofline --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FSF_BWD_SPLIT=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms')"; done; } > $out/ab_train_k10p.txt
{ hdr; echo "# the launch / host-wait work of the round's second half, all off vs all on (default), interleaved, 10-sweep and single-sweep frame"
  OFF="FSF_OVERLAP_ROWS=0 FSF_UNIQUE_BOUNDS=0 FSF_REFINE_DIRECT=0 FSF_READBACK_MAILBOX=0"
  for rep in 1 2 3; do for e in "$OFF" "FSF_OVERLAP_ROWS=1"; do for sw in 10 1; do
    v=$(env $e python bench.py --sweeps $sw --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms')")
    echo "rep $rep  ${sw}-sweep  [$( [ "$e" = "$OFF" ] && echo K26/bounds/direct-refine/mailbox OFF || echo default )]  $v"
  done; done; done; } > $out/ab_launch_tail.txt
tail -c 600 $out/bench_final.json; echo; tail -c 300 $out/bench_train.json; echo; head -5 $out/kernel_stats_full_forward.txt

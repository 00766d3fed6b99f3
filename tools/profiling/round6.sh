#!/bin/bash
# usage (on the GPU box): tools/profiling/round6.sh <commit> [light]  -> gpurun_out/r6/*  (the files copied into profiles/ as r6_*)
# Every text file starts with the commit it was taken at; the JSON lines carry it as "commit" (bench.py reads FSF_COMMIT).
commit=$1; light=$2
out=gpurun_out/r6
mkdir -p $out
export TMPDIR=/tmp FSF_COMMIT=$commit
hdr() { echo "# commit $commit, one MI355X box ($(hostname)), $(date -u +%Y-%m-%dT%H:%MZ)"; }
prof() {  # <name> <frames> <bench args...>: rocprofv3 kernel trace of a bench run -> kernel table (+ timeline of the last frame)
  name=$1; frames=$2; shift; shift
  rm -rf gpurun_out/prof_tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py "$@" > $out/bench_under_rocprof_$name.json 2>> $out/bench.err
  { hdr; python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db $frames "rocprofv3 --kernel-trace --stats -- python bench.py $*"; } > $out/kernel_stats_$name.txt
  if [ "$name" != "train_step" ]; then { hdr; python tools/profiling/frame_timeline.py gpurun_out/prof_tmp/fsf_results.db $frames; } > $out/frame_timeline_$name.txt; fi
  rm -rf gpurun_out/prof_tmp
}
FWD="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --no-h2d"
prof full_forward 7 $FWD
prof full_forward_serial 7 $FWD --serial
prof config2_1sweep 7 $FWD --sweeps 1
cp $out/kernel_stats_full_forward_serial.txt profiles/r6_kernel_stats_full_forward_serial.txt
cp $out/kernel_stats_full_forward.txt profiles/r6_kernel_stats_full_forward.txt
{ hdr; python tools/profiling/sync_sites.py 2>/dev/null; } > $out/host_sync_sites.txt
{ hdr; python tools/profiling/aten_sites.py 90 2>/dev/null; } > $out/aten_sites.txt
{ hdr; python tools/profiling/stage_times.py 2>/dev/null; } > $out/stage_times.txt
{ hdr; python tools/profiling/host_device_timeline.py 10 2>/dev/null | grep -v Warning | head -40; } > $out/host_device_timeline_full_forward.txt
{ hdr; python tools/profiling/host_device_timeline.py 1 2>/dev/null | grep -v Warning | head -40; } > $out/host_device_timeline_config2_1sweep.txt
{ hdr; python tools/profiling/neck_stall_probe.py 2>/dev/null | grep idle; } > $out/neck_stall_probe.txt
if [ "$light" != "light" ]; then
  prof train_step 7 --train --no-roofline --steps 5 --warmup 2
  bash tools/profiling/pmc_traffic.sh r6 > /dev/null 2>&1
  cp gpurun_out/r6_pmc_traffic.json $out/pmc_traffic.json
  cp gpurun_out/r6_pmc_traffic.json profiles/r6_pmc_traffic.json
  python bench.py > $out/bench_final.json 2>> $out/bench.err
  python bench.py --sweeps 1 --no-cpu-baseline --no-train-block > $out/bench_config2_1sweep.json 2>> $out/bench.err
  python bench.py --train --steps 10 --warmup 3 > $out/bench_train.json 2>> $out/bench.err
  python bench.py --train --frames-per-gpu 2 --steps 6 --warmup 2 > $out/bench_train_bs2.json 2>> $out/bench.err
  python bench.py --train --dataset av2 --steps 10 --warmup 3 > $out/bench_train_av2.json 2>> $out/bench.err
  python bench.py --dataset av2 --no-cpu-baseline > $out/bench_av2.json 2>> $out/bench.err
  python bench.py --trained-like --no-cpu-baseline > $out/bench_trained_like.json 2>> $out/bench.err
  { hdr; python tools/profiling/planes_layers.py 2>/dev/null; } > $out/spconv_layers_k9b_vs_k9d.txt
  { hdr; python tools/profiling/k22_calls.py 2>/dev/null; } > $out/k22_family_calls.txt
  { hdr; python tools/profiling/train_ops.py 60 2>/dev/null; } > $out/train_step_ops.txt
fi
{ hdr; python tools/profiling/frame_front_ab.py 10 7 2>/dev/null | grep -v amdgpu; python tools/profiling/frame_front_ab.py 1 7 2>/dev/null | grep -v amdgpu; } > $out/frame_front_ab.txt
{ hdr; python tools/profiling/frame_front_host.py 10 2>/dev/null | grep -v amdgpu; } > $out/frame_front_host.txt
{ hdr; python tools/profiling/host_gaps.py 2>/dev/null | grep -v amdgpu; } > $out/host_gaps_full_forward.txt
tail -c 400 $out/bench_final.json 2>/dev/null; echo; head -12 $out/kernel_stats_full_forward_serial.txt

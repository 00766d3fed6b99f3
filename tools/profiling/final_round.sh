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
python bench.py --trained-like --no-cpu-baseline > $out/bench_trained_like.json 2>> $out/bench.err
bash tools/profiling/run_prof.sh fwd_$tag > /dev/null 2>&1
cp gpurun_out/fwd_${tag}_kernels.txt $out/kernel_stats_full_forward.txt
cp gpurun_out/fwd_${tag}_bench.json $out/bench_under_rocprof.json
FSF_UNET_LATERAL_STREAM=0 bash tools/profiling/run_prof.sh fwdserial_$tag > /dev/null 2>&1   # the setting bench.py's conv events are taken in
cp gpurun_out/fwdserial_${tag}_kernels.txt $out/kernel_stats_full_forward_serial_unet.txt
rm -rf gpurun_out/prof_tr
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tr -o fsf -- python bench.py --train --no-roofline --steps 5 --warmup 2 > $out/bench_train_under_rocprof.json 2>> $out/bench.err
python tools/profiling/prof_summary.py gpurun_out/prof_tr/fsf_results.db 7 "rocprofv3 --kernel-trace --stats -- python bench.py --train --no-roofline --steps 5 --warmup 2 ($tag)" > $out/kernel_stats_train_step.txt
rm -rf gpurun_out/prof_tr
bash tools/profiling/pmc_traffic.sh $tag > /dev/null 2>&1
cp gpurun_out/${tag}_pmc_traffic.json $out/pmc_traffic.json
python tools/profiling/planes_layers.py > $out/spconv_layers_k9b_vs_k9d.txt 2>/dev/null
{
  echo "# rocprofv3 --pmc <counters> --kernel-trace (separate passes), averages per launch of fsf::spconv_fwd_pipe_kernel (K9d) on the"
  echo "# 101 119-row 128 -> 128 submanifold layer of the 10-sweep frame (tools/profiling/planes_one.py 2); wave counters in quad-cycles"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    bash tools/profiling/pmc_kernel.sh spconv_fwd_pipe $set -- python tools/profiling/planes_one.py 2 2>/dev/null
  done
} > $out/pmc_stalls_k9d.txt
{
  echo "# K9c (FSF_PLANES_PIPE=0) vs K9d (default) vs K9e (FSF_PLANES_WIDE_MIN_ROWS=30000) vs K9f (FSF_PLANES_TRI_MIN_ROWS=1) vs K9g (FSF_PLANES_R96_MIN_ROWS=1), same box, tools/profiling/planes_one.py"
  for e in "FSF_PLANES_PIPE=0" "FSF_PLANES_PIPE=1" "FSF_PLANES_WIDE_MIN_ROWS=30000" "FSF_PLANES_TRI_MIN_ROWS=1" "FSF_PLANES_R96_MIN_ROWS=1"; do echo "[$e]"; env $e python tools/profiling/planes_one.py 0 1 2 4 10 12 22 23 24 2>/dev/null | tr "|" "\n"; done
} > $out/spconv_k9c_k9d_k9e.txt
bash tools/profiling/pipe_ablate.sh > /dev/null 2>&1
cp gpurun_out/pipe_ablate.txt $out/spconv_k9d_ablations.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/gprobe tools/profiling/gather_pattern_probe.hip 2>/dev/null && { /tmp/gprobe 8192; /tmp/gprobe 200000; } > $out/vmem_return_probe.txt
python tools/profiling/planes_tail_probe.py 2 4 10 > $out/spconv_tail_probe.txt 2>/dev/null
bash tools/profiling/pmc_k22.sh > /dev/null 2>&1
cp gpurun_out/pmc_k22.txt $out/pmc_k22_vs_k22b.txt
bash tools/profiling/lna_ablate.sh > /dev/null 2>&1
cp gpurun_out/lna_ablate.txt $out/k22_ablations.txt
{ for p in 0 1; do echo "[FSF_K22_F16=$p]"; FSF_K22_F16=$p python tools/profiling/lna_bench.py 2>/dev/null | sed "s/F.linear .*//"; done; } > $out/k22_vs_k22b.txt
python tools/profiling/aten_sites.py 90 2>/dev/null > $out/aten_sites.txt
python tools/profiling/plan_stream_trace.py 2>/dev/null > $out/unet_plan_stream_trace.txt
python tools/profiling/plan_stream_ab.py 2>/dev/null >> $out/unet_plan_stream_trace.txt
python tools/profiling/tail_times.py 2>/dev/null > $out/box_tail_times.txt
FSF_BOX_TAIL_FUSED=0 python tools/profiling/tail_times.py 2>/dev/null | sed "s/^/[FSF_BOX_TAIL_FUSED=0] /" >> $out/box_tail_times.txt
python tools/profiling/stage_times.py 2>/dev/null > $out/stage_times.txt
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tl -o fsf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like > /dev/null 2>&1
python tools/profiling/frame_timeline.py gpurun_out/prof_tl/fsf_results.db 7 > $out/frame_timeline.txt
python tools/profiling/dispatch_list.py gpurun_out/prof_tl/fsf_results.db 7 seg_reduce_kernel seg_fixup_long sir_input_kernel linear_norm_act_kernel > $out/dispatches_in_situ.txt
rm -rf gpurun_out/prof_tl
bash tools/profiling/ab_bench.sh "FSF_UNET_PLAN_STREAM=0 FSF_BOX_TAIL_FUSED=0" "FSF_UNET_PLAN_STREAM=1 FSF_BOX_TAIL_FUSED=0" "FSF_UNET_PLAN_STREAM=1 FSF_BOX_TAIL_FUSED=1" > $out/ab_plan_stream_box_tail.txt 2>/dev/null
tail -c 700 $out/bench.json; echo; tail -c 300 $out/bench_train.json; echo; tail -c 300 $out/bench_train_bs2.json; echo; tail -c 300 $out/bench_av2.json

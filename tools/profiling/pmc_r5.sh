#!/bin/bash
# usage (GPU box): tools/profiling/pmc_r5.sh <commit> -> gpurun_out/r5/pmc_*.txt : SQ / TCC counters (rocprofv3 --pmc + --kernel-trace only, one
# counter set per pass) of the kernels this round's work is about: K22s, K22h (and the same product on K22), K9d, K10p
commit=$1; out=gpurun_out/r5; mkdir -p $out
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum")
run() {  # <file> <title> <kernel substring> <cmd...>
  f=$out/$1; title=$2; pat=$3; shift; shift; shift
  { echo "# commit $commit, one MI355X box ($(hostname)), $(date -u +%Y-%m-%dT%H:%MZ)"
    echo "# rocprofv3 --pmc <one set per pass> --kernel-trace -- $*   (averages per launch of kernels matching '$pat'; wave counters in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE in cycles)"
    echo "# $title"; } > $f
  for set in "${SETS[@]}"; do bash tools/profiling/pmc_kernel.sh "$pat" $set -- "$@" 2>/dev/null >> $f; done
}
run pmc_k22s.txt "K22s: fsf_linear_norm_act_segmax, 510 652 rows sorted by group x 128 -> 128, grouped (row_add), LayerNorm + GELU + segmented max, rows written" "linear_norm_act_kernel<8, 4, true" python tools/profiling/k22s_one.py 5
run pmc_k22h.txt "K22h: fsf_linear_planes_norm_act, 10 641 rows x 1024 -> 1024 + bias, f16 x 3 planes" "linear_norm_act_kernel<8, 4, false, -1, -1, 1>" python tools/profiling/k22h_one.py 5
FSF_K22H_ONE=k22 run pmc_k22_wide_bf16x6.txt "the same product on K22 (bf16 x 6, x split in the kernel, 8 slices on blockIdx.y)" "linear_norm_act_kernel<8, 4, false, -1, -1, 0>" python tools/profiling/k22h_one.py 5
run pmc_stalls_k9d.txt "K9d: fsf::spconv_fwd_pipe_kernel on the 101 119-row 128 -> 128 submanifold layer (tools/profiling/planes_one.py 2)" "spconv_fwd_pipe_kernel" python tools/profiling/planes_one.py 2
run pmc_k10p.txt "K10p: fsf::spconv_bwd_weight_split_kernel over the frame's 34 layers (tools/profiling/bwd_weight_layers.py)" "spconv_bwd_weight_split_kernel" python tools/profiling/bwd_weight_layers.py

#!/bin/bash
# usage (GPU box): tools/profiling/pmc_k22.sh -> gpurun_out/pmc_k22.txt : SQ counters of K22 (bf16 x6) and K22b (f16 x3, coalesced x) on the 510 k x 256 -> 128 call
out=gpurun_out/pmc_k22.txt; : > $out
for f16 in 0 1; do
  echo "# FSF_K22_F16=$f16 (rocprofv3 --pmc ... --kernel-trace, separate passes; averages per launch; wave counters in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles)" >> $out
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
    FSF_K22_F16=$f16 bash tools/profiling/pmc_kernel.sh linear_norm_act $set -- python tools/profiling/lna_one.py 2>/dev/null >> $out
  done
done
cat $out

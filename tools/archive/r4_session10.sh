#!/bin/bash
# Round-4 GPU session 10 (~2 GPU-minutes, evidence only -- no product code involved): SQ counters of the MFMA kernels.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s10; mkdir -p $O
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES"; do
  d=/tmp/pmc_$(echo $pass | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -o k -- python $GRAFT_REPO_ROOT/tools/r4_pmc_kernels.py > $d/run.log 2>&1)
  tail -1 $d/run.log
done
python tools/pmc_by_kernel.py $O/r4_mfma_kernels_pmc.json /tmp/pmc_SQ_WAVE_CYCLES /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES /tmp/pmc_SQ_INSTS_VALU --match "flash_attn|gemm_8phase" 2>&1 | tail -150

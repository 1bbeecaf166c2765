#!/bin/bash
# Round-6 GPU session 16 (~12 GPU-minutes): PMC passes on the tree with the fusions -- (a) one 40-row + one 12-row eager UNet forward (the
# launch mix of bench.py's default: two images in flight) -> profiles/r6_unet_pmc.json (bench.py reads roofline.traffic from it), (b) the MFMA
# kernels at their largest shapes -> profiles/r6_final_mfma_kernels_pmc.json.  Counter passes run with --kernel-trace only.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s16; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  d=/tmp/pmc_$(echo $c | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o unet -- python $GRAFT_REPO_ROOT/tools/pmc_unet.py 40,12 > $d/run.log 2>&1)
  tail -1 $d/run.log
done
python tools/pmc_summarise.py --rows 40,12 $O/r6_unet_pmc.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $O/pmc_summarise.log 2>&1; tail -3 $O/pmc_summarise.log
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmck_$(echo $pass | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -o k -- python $GRAFT_REPO_ROOT/tools/r6_pmc_kernels.py > $d/run.log 2>&1)
  tail -1 $d/run.log
done
python tools/pmc_by_kernel.py $O/r6_final_mfma_kernels_pmc.json /tmp/pmck_SQ_WAVE_CYCLES /tmp/pmck_SQ_VALU_MFMA_BUSY_CYCLES /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE --match "flash_attn|gemm_8phase|geglu_persist|gn32_nhwc|gn_nhwc" > $O/pmc_by_kernel.log 2>&1; tail -3 $O/pmc_by_kernel.log
du -sh $O

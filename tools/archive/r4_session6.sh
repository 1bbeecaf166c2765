#!/bin/bash
# Round-4 GPU session 6 (~9 GPU-minutes): evidence for profiles/ -- attention ablations, rocprofv3 --kernel-trace --stats of
# the bench command, PMC passes over the UNet's hand-written kernels.  Big rocprof files stay in /tmp on the box.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s6; mkdir -p $O
( time timeout 200 python tools/attn_ablate/run.py --out $O/attn_ablation.jsonl ) > $O/attn_ablation.log 2>&1
cat $O/attn_ablation.jsonl
P=/tmp/prof_bench; mkdir -p $P
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $P -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -14 $O/trace_summary.txt
head -16 $O/bench_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  d=/tmp/pmc_$(echo $c | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o unet -- python $GRAFT_REPO_ROOT/tools/pmc_unet.py > $d/run.log 2>&1)
  tail -1 $d/run.log
done
python tools/pmc_summarise.py $O/r4_unet_pmc.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | tail -60
du -sh $O

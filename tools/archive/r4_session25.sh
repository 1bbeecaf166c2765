#!/bin/bash
# Round-4 GPU session 25 (<0.5 GPU-minute, experiment only): the library built from the product sources + tools/r5_patches/*.patch against the
# product library, entry point by entry point (bit identity + timing).  The product tree itself is unchanged.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s25; mkdir -p $O
( time timeout 80 python tools/r5_patches/probe_patched.py --rounds 3 ) > $O/patched_vs_product.jsonl 2> $O/patched_vs_product.err
cat $O/patched_vs_product.jsonl; tail -3 $O/patched_vs_product.err

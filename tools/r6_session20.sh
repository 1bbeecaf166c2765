#!/bin/bash
# Round-6 GPU session 20 (~4 GPU-minutes): the VAE encoder's downsamplers on the split-operand path: parity tests, pad-strip encode A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s20; mkdir -p $O
( time timeout 900 python -m pytest tests/test_vae_split.py -m gpu -x -q ) > $O/pytest_vae.log 2>&1; tail -4 $O/pytest_vae.log
timeout 300 python tools/r5_vae_ab.py --cases encode --reps 7 --switch VAE_SPLIT_DOWNSAMPLE > $O/vae_downsample_ab.jsonl 2> $O/vae_ab.err; cat $O/vae_downsample_ab.jsonl | cut -c1-1200; tail -2 $O/vae_ab.err

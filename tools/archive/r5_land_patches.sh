#!/bin/bash
# Round 5: every patch of tools/r5_patches has been landed in the product tree (tools/r5_patches/landed/ keeps them as the record of
# what was prepared at the end of round 4): 0001-0005 as they were (session 1: bit-identical to the round-4 library on 54 cases, GPU
# suite green, forward 145.6 -> 143.0 ms), 0006 -- the long-K loop -- merged by hand for the CONVOLUTIONS only (session 2: +4...12 %
# there, -0.5...-2.7 % on the K >= 1280 projections).  Nothing left to apply.
echo "all round-5 patches are in the product tree (see tools/r5_patches/README.md)"

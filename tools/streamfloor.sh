#!/bin/bash
# tools/streamfloor.hip over launch geometries; run on the GPU box.
cd $GRAFT_REPO_ROOT
B=./build/streamfloor
for wg in 256 512 1024; do for u in 2 4 6 8 12 16; do $B 398 $wg $u 1; done; done
for wg in 256 512; do for u in 4 8; do for t in 2 3 4 8; do $B 398 $wg $u $t; done; done; done
for mb in 100 200 800 1600 4000; do $B $mb 256 6 1; done

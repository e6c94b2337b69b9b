#!/bin/bash
# tools/scanfloor.hip over the shapes that matter; run on the GPU box.  Arguments:
#   <elems/block> <rows/block> <nt> <y> <meta> <compute> <ipt> <descriptor> <gather flavour> <p_local>
cd $GRAFT_REPO_ROOT
B=./build/scanfloor
if [ "$1" != "gathers" ]; then
$B 2047 682 1 1 1 0 8      # the SCAN block shape: 2047 nonzeros, ~682 rows
$B 2047 682 1 0 1 0 8      # no y
$B 2047 682 1 1 0 0 8      # no meta
$B 2048 682 1 1 1 0 8      # line-aligned blocks
$B 2047 682 0 1 1 0 8      # plain (cached) loads
$B 2047 682 1 1 1 1 8      # + run sums
$B 2047 682 1 1 1 2 8      # + wave scan
$B 4095 1364 1 1 1 0 16    # 16 per thread
$B 1023 341 1 1 1 0 4      # 4 per thread
$B 2047 32 1 1 1 0 8       # cant-like rows per block
$B 2047 682 1 1 1 2 8 1    # the first shape + compute, block start from a cold descriptor
$B 2047 682 1 1 1 0 8 1
$B 4095 1364 1 1 1 0 16 1
fi
for g in 1 2 3 4 5 7 8 6; do $B 2047 682 1 1 1 2 8 1 $g 0.7; done     # + webbase-like gathers, by load flavour
for g in 1 4 5; do $B 2047 682 1 1 1 2 8 1 $g 1.0; done          # all local
for g in 1 4 5; do $B 2047 682 1 1 1 2 8 1 $g 0.0; done          # all scattered
for p in 0.9 0.8 0.6 0.5; do $B 2047 682 1 1 1 2 8 1 1 $p; done
$B 1023 341 1 1 1 2 4 1 1 0.7
$B 4095 1364 1 1 1 2 16 1 1 0.7

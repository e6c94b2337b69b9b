#!/bin/bash
# Variant SLICE against SCAN on the power-law look-alikes (cold us per launch, every row checked by tests elsewhere):
# the gate of VERDICT r5 item 1 -- webbase2 <= 13.0 us / webbase-1M <= 19.5 us or the A/B on record.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/slice_probe.txt
: > $OUT
for m in ${@:-webbase2 webbase-1M}; do
  timeout -k 10 500 python3 tools/family_times.py $m \
    'variant=scan,wg_size=256,items_per_thread=8,tile_width=2048' \
    'variant=slice,lanes_per_row=4,wg_size=256,items_per_thread=8,tile_width=2048' \
    'variant=slice,lanes_per_row=2,wg_size=256,items_per_thread=8,tile_width=2048' \
    'variant=slice,lanes_per_row=1,wg_size=256,items_per_thread=8,tile_width=2048' \
    'variant=slice,lanes_per_row=3,wg_size=256,items_per_thread=8,tile_width=2048' \
    'variant=slice,lanes_per_row=8,wg_size=256,items_per_thread=8,tile_width=2048' \
    'variant=slice,lanes_per_row=4,wg_size=256,items_per_thread=8,tile_width=-1' \
    'variant=slice,lanes_per_row=4,wg_size=256,items_per_thread=4,tile_width=1024' \
    'variant=slice,lanes_per_row=4,wg_size=512,items_per_thread=4,tile_width=2048' \
    'variant=slice,lanes_per_row=4,wg_size=128,items_per_thread=8,tile_width=1024' \
    'variant=slice,lanes_per_row=2,wg_size=512,items_per_thread=8,tile_width=4096' \
    'variant=scan,wg_size=512,items_per_thread=4,tile_width=2048' \
    'variant=vector,lanes_per_row=2' 'variant=vector,lanes_per_row=4' 'variant=vector,lanes_per_row=1' \
    2>gpurun_out/slice_probe.err | cut -c1-330 >> $OUT || { tail -5 gpurun_out/slice_probe.err; exit 1; }
done
cat $OUT

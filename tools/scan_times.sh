#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/family_times.py cant 'variant=merge,tile_width=1024' 'variant=merge_pair,tile_width=1024' 'variant=merge,tile_width=1024' 'variant=merge_pair,tile_width=1024' 'variant=merge_pair,tile_width=1024,wg_size=512' 2>&1 | cut -c1-150
python tools/family_times.py G3_circuit 'variant=merge' 'variant=merge_pair' 2>&1 | cut -c1-150
python tools/family_times.py atmosmodd 'variant=merge' 'variant=merge_pair' 2>&1 | cut -c1-150

#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/family_times.py webbase-1M 'variant=scan,tile_width=4096,far_columns=-1' 'variant=scan,tile_width=-1,far_columns=2' 'variant=scan,tile_width=-1,far_columns=3' 'variant=scan,tile_width=4096,far_columns=3' 'variant=scan,tile_width=-1,far_columns=3,items_per_thread=4' 'variant=scan,tile_width=-1,far_columns=3,wg_size=512' 2>&1 | cut -c1-140
python tools/family_times.py G3_circuit 'variant=scan,tile_width=-1,far_columns=-1'  'variant=scan,tile_width=-1,far_columns=3' 2>&1 | cut -c1-130

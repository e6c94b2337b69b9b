#!/bin/bash
cd $GRAFT_REPO_ROOT
specs=""
for wg in 64 128 256 512; do for it in 4 8 16; do specs="$specs variant=merge,tile_width=1024,wg_size=$wg,items_per_thread=$it"; done; done
python tools/family_times.py cant $specs 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); p=r['resolved']; print(p['wg_size'], p['items_per_thread'], p['tile_width'], p['index16'], r['usec_cold'], r['grid'], r['lds'])
    else: print(l.strip()[:200])
"

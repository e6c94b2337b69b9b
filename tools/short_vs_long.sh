#!/bin/bash
# the driver's short region (--steps 20 --warmup 5) against a long one on the same box
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('steps 20  ', r['ms_per_step']*1e3, r['value'], r['config']['launch'], r['config']['untimed_preroll_replays'], r['config']['tune']['seconds'])"
done
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('steps 1000', r['ms_per_step']*1e3, r['value'], r['config']['launch'], r['config']['untimed_preroll_replays'])"
( time python bench.py --steps 20 --warmup 5 > gpurun_out/driver_form.json 2>/dev/null ) 2>&1 | grep real

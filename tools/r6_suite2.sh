#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/suite_time.sh r06 > gpurun_out/suite_time.txt 2>&1; cat gpurun_out/suite_time.txt | tail -45
timeout -k 10 300 python3 tools/family_times.py webbase2 'variant=vector,lanes_per_row=2' 'variant=vector,lanes_per_row=1' 'variant=vector,lanes_per_row=4' 'variant=vector,lanes_per_row=8' > gpurun_out/vector_times.txt 2>/dev/null; cut -c1-140 gpurun_out/vector_times.txt
timeout -k 10 300 python3 tools/family_times.py webbase-1M 'variant=vector,lanes_per_row=2' 'variant=vector,lanes_per_row=1' >> gpurun_out/vector_times.txt 2>/dev/null; tail -2 gpurun_out/vector_times.txt | cut -c1-140
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-others --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['host_entry'])"

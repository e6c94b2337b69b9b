#!/bin/bash
# Row f4 next to the reference's CPU path, one box:  tools/trsv_round.sh <tag>   ->  gpurun_out/<tag>_trsv.jsonl (default
# schedules) and gpurun_out/<tag>_trsv_walk2.jsonl (CASK_HIP_TRSV=walk2: the A/B of the lane-group walk, cant only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=${1:-x}
timeout -k 10 700 python tools/bench_solvers.py trsv G3_circuit cant atmosmodd 2>/dev/null | grep '^{' > gpurun_out/${tag}_trsv.jsonl; echo "default rc=$?"
CASK_HIP_TRSV=walk2 timeout -k 10 300 python tools/bench_solvers.py trsv cant 2>/dev/null | grep '^{' > gpurun_out/${tag}_trsv_walk2.jsonl; echo "walk2 rc=$?"
CASK_HIP_TRSV_STATS=1 timeout -k 10 120 python tools/trsv_bench.py cant ilu0_unit 2>&1 | grep "cycles/chunk\|^lanes" | tail -4 > gpurun_out/${tag}_trsv_stats.txt
python - <<'PY' ${tag}
import json, sys
tag = sys.argv[1]
for f in (f"gpurun_out/{tag}_trsv.jsonl", f"gpurun_out/{tag}_trsv_walk2.jsonl"):
    for line in open(f):
        r = json.loads(line)
        cpu = r["cpu"]["unit_lower"]
        print(f, r["matrix"], "gpu", r["gpu"]["ms_per_application"], "mkl", {k: v["ms_per_application"] for k, v in cpu.items()},
              "gpu passes", r.get("gpu_passes"))
PY
cat gpurun_out/${tag}_trsv_stats.txt

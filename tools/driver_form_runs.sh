#!/bin/bash
# The driver's exact command, N times in a row on this lease; one JSON line per run appended to
# gpurun_out/driver_form_<tag>.jsonl (tools/driver_form_summary.py folds the leases into profiles/r04_driver_form_runs.json).
#   tools/driver_form_runs.sh <tag> [N]
set -o pipefail
tag=${1:-lease}; n=${2:-4}
mkdir -p gpurun_out
out=gpurun_out/driver_form_${tag}.jsonl
: > "$out"
for i in $(seq 1 "$n"); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/driver_form_${tag}_$i.err | tail -n 1 >> "$out" || exit 1
  python3 - "$out" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("run", r["ms_per_step"] * 1e3, "us  p10/p90", r.get("ms_per_step_p10", 0) * 1e3, r.get("ms_per_step_p90", 0) * 1e3,
      "first", r.get("ms_per_step_first_window", 0) * 1e3, "frac", r["roofline"]["frac"], r.get("gpu_clocks_mhz"), flush=True)
PY
done

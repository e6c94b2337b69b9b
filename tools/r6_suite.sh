#!/bin/bash
# the whole GPU suite as the driver runs it + the driver's bench command (one lease)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/suite.log 2>&1; RC=$?
T1=$(date +%s)
echo "pytest -m gpu: exit $RC, $((T1 - T0)) s wall" | tee gpurun_out/suite_time.txt
tail -30 gpurun_out/suite.log
[ $RC -ne 0 ] && exit 1
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print(r["value"], r["unit"], r["ms_per_step"], "frac", r["roofline"]["frac"], "host_entry", r.get("host_entry"), "traffic", r["roofline"]["traffic"])
for o in r["config"].get("other_workloads", []):
    print("  ", o.get("config"), o.get("workload"), o.get("usec"), o.get("frac"), o.get("design_point"), o.get("rows_wrong"), (o.get("solve_check") or {}).get("iterations"), o.get("traffic"), str(o.get("traffic_source"))[:90], o.get("error"))
PY

#!/bin/bash
# Round-6 closing evidence in one lease: the bench line as the driver runs it + its rocprofv3 kernel statistics + the headline
# point's traffic (tools/profile_round.sh).  The DSE with counters and the other traffic files: tools/r6_extra_pmc.sh (its own
# lease: profile_round deletes the calibration's counter files).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/profile_round.sh r06 cant > gpurun_out/profile_round_r06.log 2>&1; tail -2 gpurun_out/profile_round_r06.log | cut -c1-400
bash tools/pmc_solver.sh r06 G3_circuit cg > /dev/null 2>&1
bash tools/pmc_solver.sh r06 atmosmodd bicg > /dev/null 2>&1
python3 - <<'PY'
import json
for f in ("gpurun_out/traffic_G3_circuit_cg_r06.json", "gpurun_out/traffic_atmosmodd_bicg_r06.json"):
    t = json.load(open(f)); print(f, t.get("hbm_bytes_per_launch"), {k: v.get("bytes_per_launch") for k, v in t["kernels"].items()})
PY

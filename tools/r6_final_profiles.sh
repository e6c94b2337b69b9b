#!/bin/bash
# Round-6 evidence in one lease: the bench line as the driver runs it + its rocprofv3 kernel statistics + the AUTO point's
# traffic (tools/profile_round.sh), then the DSE over the five matrices with counter evidence per winner and runner-up.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/profile_round.sh r06 cant > gpurun_out/profile_round_r06.log 2>&1; tail -4 gpurun_out/profile_round_r06.log | cut -c1-400
timeout -k 10 1500 python3 tools/dse_evidence.py r06 gpurun_out/dse_out_r06.json cant G3_circuit webbase-1M webbase2 atmosmodd 2>&1 | grep -v "amdgpu.ids" | tail -12
ls gpurun_out | grep -c traffic
du -sh gpurun_out

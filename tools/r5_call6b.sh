#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python3 tools/dse_evidence.py r05 gpurun_out/r05_dse_out.json cant G3_circuit webbase-1M webbase2 atmosmodd > gpurun_out/r05_dse_evidence.log 2>&1; echo "dse_evidence rc=$?"; tail -8 gpurun_out/r05_dse_evidence.log | cut -c1-300
bash tools/pmc_solver.sh r05 G3_circuit cg > gpurun_out/r05_pmc_cg.log 2>&1; echo "pmc cg rc=$?"; tail -3 gpurun_out/r05_pmc_cg.log | cut -c1-300
bash tools/pmc_solver.sh r05 atmosmodd bicg > gpurun_out/r05_pmc_bicg.log 2>&1; echo "pmc bicg rc=$?"; tail -3 gpurun_out/r05_pmc_bicg.log | cut -c1-300
ls gpurun_out | grep -i "traffic.*r05\|r05_dse"

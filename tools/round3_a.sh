#!/bin/bash
# round 3, first measurement pass: full GPU suite, DSE over the four families, traffic of the webbase-like winners
cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1; echo tests rc=$?; tail -4 gpurun_out/t_full.log
timeout -k 10 600 python tools/dse.py --out gpurun_out/r03_dse_out.json webbase-1M G3_circuit atmosmodd cant > gpurun_out/r03_dse.log 2>&1; grep best gpurun_out/r03_dse.log | cut -c1-200
timeout -k 10 300 bash tools/pmc_point.sh r03 webbase-1M scan4096 --variant scan --tile 4096 --far -1 2>&1 | tail -30
timeout -k 10 300 bash tools/pmc_point.sh r03 webbase-1M scanfar2 --variant scan --tile -1 --far 2 2>&1 | grep -A3 "hbm_bytes\|read_requests" | head -20

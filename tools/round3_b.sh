#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1; echo tests rc=$?; tail -4 gpurun_out/t_full.log
timeout -k 10 900 python tools/dse_evidence.py r03 gpurun_out/r03_dse_out.json webbase-1M G3_circuit atmosmodd cant 2>&1 | grep -v amdgpu.ids | tail -12

#!/bin/bash
# counter evidence of round 6 in one lease (no tools/profile_round.sh in front: it deletes the calibration's counter files):
# the DSE over the five matrices with three PMC passes per winner (and runner-up within 1 %), then the design points the default
# bench line times besides (cant3, the 512 x 8 SCAN shapes, the unpadded SCAN plan next to the padded one) and the solver passes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1500 python3 tools/dse_evidence.py r06 gpurun_out/dse_out_r06.json cant G3_circuit webbase-1M webbase2 atmosmodd 2>&1 | grep -v "amdgpu.ids" | tail -6
CASK_HIP_SCAN_PAD=0 bash tools/pmc_point.sh r06 webbase2 scan_w256_i8_t2048_l0_nopad --variant scan --wg 256 --items 8 --tile 2048 > /dev/null 2>&1
bash tools/pmc_point.sh r06 webbase2 scan_w512_i8_t2048_l0 --variant scan --wg 512 --items 8 --tile 2048 > /dev/null 2>&1
bash tools/pmc_point.sh r06 webbase-1M scan_w512_i8_t4096_l0 --variant scan --wg 512 --items 8 --tile 4096 > /dev/null 2>&1
bash tools/pmc_point.sh r06 cant3 merge_w256_i8_t512_l16 --variant merge --wg 256 --items 8 --tile 512 --lanes 16 > /dev/null 2>&1
bash tools/pmc_point.sh r06 cant3 merge_w256_i8_t1024_l16 --variant merge --wg 256 --items 8 --tile 1024 --lanes 16 > /dev/null 2>&1
bash tools/pmc_point.sh r06 atmosmodd merge_w512_i8_t2048_l2 --variant merge --wg 512 --items 8 --tile 2048 --lanes 2 > /dev/null 2>&1
bash tools/pmc_solver.sh r06 G3_circuit cg > /dev/null 2>&1
bash tools/pmc_solver.sh r06 atmosmodd bicg > /dev/null 2>&1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/traffic_*_r06.json")):
    t = json.load(open(f))
    print(f.split("/")[-1], t.get("kernel"), "fetch KiB", t.get("FETCH_SIZE_KiB"), "write KiB", t.get("WRITE_SIZE_KiB"), "bytes", t.get("hbm_bytes_per_launch") or t.get("hbm_bytes_per_pass"))
PY

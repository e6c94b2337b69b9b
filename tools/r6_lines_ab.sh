#!/bin/bash
# r6: interleaved A/B, 16-column line tiles (the product library) against 64-column chunk tiles (the previous build,
# build/diag/libcask_hip_chunks64.so), one box: products at fixed design points, then the solver passes.
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
old=build/diag/libcask_hip_chunks64.so
bash tools/ab_lib.sh lines_g3 $old --workload G3_circuit --variant merge --wg 256 --items 8 --tile 2048 --lanes 1 --steps 400 --warmup 50 || exit 1
bash tools/ab_lib.sh lines_at256 $old --workload atmosmodd --variant merge --wg 256 --items 8 --tile 2048 --lanes 2 --steps 400 --warmup 50 || exit 1
bash tools/ab_lib.sh lines_at512 $old --workload atmosmodd --variant merge --wg 512 --items 8 --tile 2048 --lanes 2 --steps 400 --warmup 50 || exit 1
bash tools/ab_solver.sh lines_cg $old G3_circuit cg || exit 1
bash tools/ab_solver.sh lines_bicg $old atmosmodd bicg || exit 1

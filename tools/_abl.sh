cd $GRAFT_REPO_ROOT
for a in 0 32 33; do echo "abl $a"; W2_ABL=$a CASK_HIP_TRSV=walk2 python tools/trsv_bench.py G3_circuit 2>/dev/null | cut -c70-130; done

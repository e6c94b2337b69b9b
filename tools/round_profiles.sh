#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call:  tools/round_profiles.sh <tag>
set -u
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
bash $root/tools/profile_round.sh $tag cant webbase-1M G3_circuit atmosmodd > $out/profile_round_$tag.log 2>&1
tail -3 $out/profile_round_$tag.log | cut -c1-300
cd $root
python3 tools/bench_solvers.py > $out/solvers_$tag.json 2> $out/solvers_$tag.err
cat $out/solvers_$tag.json
python3 tools/dse.py --out $out/dse_out_$tag.json cant G3_circuit webbase-1M atmosmodd > $out/dse_$tag.log 2>&1 || tail -5 $out/dse_$tag.log
tail -8 $out/dse_$tag.log | cut -c1-200
du -sh $out

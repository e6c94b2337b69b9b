#!/bin/bash
# r6: line-granular chunk tiles (plan::build_chunk_tiles) -- parity of everything that runs the merge kernels, then the
# G3_circuit / atmosmodd products and solver passes and the counter traffic at the G3 design point.  Run on the GPU box.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; mkdir -p $out; cd $root
tag=${1:-r06lines}
timeout -k 10 600 python -m pytest tests/test_spmv_gpu.py tests/test_random_gpu.py tests/test_p2p_gpu.py tests/test_solvers_gpu.py tests/test_fused_gpu.py tests/test_nonfinite_gpu.py -x -q -m gpu > $out/${tag}_tests.log 2>&1 || { tail -30 $out/${tag}_tests.log; exit 1; }
tail -2 $out/${tag}_tests.log
for w in G3_circuit atmosmodd cant; do
  timeout -k 10 300 python bench.py --workload $w --no-others --no-cpu-baseline --steps 400 --warmup 50 > $out/${tag}_$w.json 2> $out/${tag}_$w.err || exit 1
  python - $out/${tag}_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(c["workload"], round(d["ms_per_step"]*1e3,3),"us", d["roofline"]["frac"], c.get("design_point") or {k:c.get(k) for k in ("variant","wg_size","items_per_thread","tile_width","lanes_per_row")})
PY
done
timeout -k 10 300 python bench.py --workload G3_circuit --solver cg --no-others --no-cpu-baseline --steps 200 --warmup 20 > $out/${tag}_cg.json 2> $out/${tag}_cg.err || exit 1
timeout -k 10 300 python bench.py --workload atmosmodd --solver bicg --no-others --no-cpu-baseline --steps 200 --warmup 20 > $out/${tag}_bicg.json 2> $out/${tag}_bicg.err || exit 1
for f in cg bicg; do python - $out/${tag}_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], d["metric"], round(d["ms_per_step"]*1e3,3),"us", d["roofline"]["frac"])
PY
done
bash tools/pmc_point.sh $tag G3_circuit merge_w256_i8_t2048_l1 --variant merge --wg 256 --items 8 --tile 2048 --lanes 1 > /dev/null 2>&1
python - $out/traffic_G3_circuit_merge_w256_i8_t2048_l1_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("G3 traffic", d["hbm_bytes_per_launch"], d["FETCH_SIZE_KiB"], d["WRITE_SIZE_KiB"])
PY

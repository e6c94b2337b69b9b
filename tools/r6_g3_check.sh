#!/bin/bash
# r6: the G3_circuit-like product, line tiles (product library) against 64-column chunks (build/diag/libcask_hip_chunks64.so), three clocks:
# rocprofv3 kernel durations of the forced bench point, the bench line itself, the DSE's cold time of the same point.
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; cd /tmp && export TMPDIR=/tmp
for which in new old; do
  if [ $which = old ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/diag/libcask_hip_chunks64.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
  rm -rf $out/g3chk_$which
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/g3chk_$which -- python3 $root/bench.py --workload G3_circuit --variant merge --wg 256 --items 8 --tile 2048 --lanes 1 --steps 400 --warmup 50 --no-others --no-cpu-baseline > $out/g3chk_$which.json 2> $out/g3chk_$which.err || exit 1
  python3 - $out/g3chk_$which $which <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_spmv_merge" in r["Name"]: print(sys.argv[2], "rocprof", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"])
d=json.loads(open(sys.argv[1]+".json").read().strip().splitlines()[-1]); print(sys.argv[2], "bench line", round(d["ms_per_step"]*1e3,3), "us")
PY
  find $out/g3chk_$which -name "*.csv" ! -name "*kernel_stats.csv" -delete
  cd $root && python3 tools/dse.py --out $out/g3chk_dse_$which.json G3_circuit 2>&1 | grep "best" ; cd /tmp
done

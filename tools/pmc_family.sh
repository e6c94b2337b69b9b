#!/bin/bash
# SQ / LDS / TA counters of the product kernel for one BASELINE look-alike (run on the GPU box), one pass per group
#   tools/pmc_family.sh <matrix> [spec]
set -u
mat=$1; spec=${2:-variant=merge}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  d=$out/pmc_${mat}_$i
  rm -rf $d
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d $d -- python3 $root/tools/family_times.py $mat $spec > $d.json 2> $d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_spmv_merge" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  find $d -name "*.csv" -size +1M -delete
done

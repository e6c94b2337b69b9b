#!/bin/bash
# tools/gatherbench.hip with timing and the L2->fabric read-request counters; run on the GPU box.
cd $GRAFT_REPO_ROOT
./build/gatherbench
./build/gatherbench 1
./build/gatherbench 2
ALLOC=${1:-0}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gb_pmc -- $GRAFT_REPO_ROOT/build/gatherbench $ALLOC > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/gb_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
find $GRAFT_REPO_ROOT/gpurun_out/gb_pmc -name "*.csv" -size +1M -delete

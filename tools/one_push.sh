cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29877
CASK_BENCH_FORCE_DIST=1 CASK_BENCH_EXCHANGE=push python bench.py --workload webbase-1M --steps 400 --warmup 40 --no-cpu-baseline --no-others --no-tune 2>gpurun_out/one.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(r['ms_per_step']*1e3, r['host_wall_ms_per_step']*1e3, r['config']['launch'], r['config']['untimed_preroll_replays'])"
tail -3 gpurun_out/one.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_push -- python3 $GRAFT_REPO_ROOT/bench.py --workload webbase-1M --steps 400 --warmup 40 --no-cpu-baseline --no-others --no-tune > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
for f in glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_push/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:5]: print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3)
PY

#!/bin/bash
# Shared-device rehearsal of the driver's SCALE command: `python3 bench.py --gpus N --steps 20 --warmup 5` for every N
# given (default 2 4 6), all ranks on the box's ONE GPU (CASK_BENCH_SHARE_DEVICE=1; RCCL refuses two ranks on one
# device, so the control plane is gloo), with the per-phase wall budget bench.py prints on stderr, the wall time of the
# whole command and the peak VRAM in use (sysfs, sampled twice a second).  The pool's process guard allows at most 6
# processes on the card, so 6 is the largest world a session may rehearse; 8 is extrapolated in profiles/r06_world_dryrun.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/world_dryrun.txt
: > $OUT
export CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo
VRAM=$(ls /sys/class/drm/card*/device/mem_info_vram_used 2>/dev/null | head -1)
for N in ${@:-2 4 6}; do
  echo "=== world $N: python3 bench.py --gpus $N --steps 20 --warmup 5 (shared device)" >> $OUT
  : > gpurun_out/vram_$N.txt
  ( while true; do cat $VRAM >> gpurun_out/vram_$N.txt 2>/dev/null; sleep 0.5; done ) &
  POLL=$!
  T0=$(date +%s.%N)
  timeout -k 10 560 python3 bench.py --gpus $N --steps 20 --warmup 5 > gpurun_out/world_$N.json 2> gpurun_out/world_$N.err
  RC=$?
  T1=$(date +%s.%N)
  kill $POLL 2>/dev/null; wait $POLL 2>/dev/null
  echo "exit status $RC, wall $(python3 -c "print(round($T1 - $T0, 1))") s, peak VRAM in use $(sort -n gpurun_out/vram_$N.txt | tail -1 | awk '{printf "%.2f GB", $1/1e9}') (idle $(head -1 gpurun_out/vram_$N.txt | awk '{printf "%.2f GB", $1/1e9}'))" >> $OUT
  grep "^\[bench\]\|phase\|budget\| s  " gpurun_out/world_$N.err | grep -v "amdgpu.ids\|socket.cpp" | tail -40 >> $OUT
  python3 - $N >> $OUT <<'PY'
import json, sys
n = sys.argv[1]
ls = [l for l in open(f"gpurun_out/world_{n}.json") if l.startswith("{")]
if not ls:
    print("no JSON line"); raise SystemExit
r = json.loads(ls[-1]); c = r["config"]
print("headline:", r["n_gpus"], "ranks", r["value"], r["unit"], r["ms_per_step"], "ms/step; exchange", str(c.get("exchange"))[:60],
      "; rows wrong", c.get("rows_wrong_vs_oracle_all_ranks"))
print("config.rccl:", json.dumps(c.get("rccl")))
print("config.xgmi:", json.dumps(c.get("xgmi")))
print("exchange_selfcheck:", json.dumps(c.get("exchange_selfcheck")))
for o in c.get("other_workloads", []):
    print("  appended:", o.get("config"), "usec", o.get("usec"), "frac", o.get("frac"), "rows_wrong", o.get("rows_wrong"),
          "iterations", (o.get("solve_check") or {}).get("iterations"), "exchange", str(o.get("exchange"))[:50],
          "seconds", o.get("seconds_in_bench"), "error", o.get("error"))
PY
  [ $RC -ne 0 ] && { echo "world $N failed: stopping" >> $OUT; tail -20 gpurun_out/world_$N.err >> $OUT; break; }
done
cat $OUT

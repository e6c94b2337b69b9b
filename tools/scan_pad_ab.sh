#!/bin/bash
# A/B of the padded SCAN plan (r6: stream addresses from the block index, the descriptor out of the stream's way) against the
# unpadded plan (CASK_HIP_SCAN_PAD=0), cold us per launch, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_spmv_gpu.py tests/test_random_gpu.py tests/test_nonfinite_gpu.py -x -q -m gpu -k "scan or random or fixtures or families_small or full_size or poisoned" > gpurun_out/pad_pytest.log 2>&1 || { tail -30 gpurun_out/pad_pytest.log; exit 1; }
tail -2 gpurun_out/pad_pytest.log
OUT=gpurun_out/scan_pad_ab.txt
: > $OUT
for rep in 1 2; do
for m in webbase2 webbase-1M; do
  for arm in pad nopad; do
    if [ $arm = nopad ]; then export CASK_HIP_SCAN_PAD=0; else unset CASK_HIP_SCAN_PAD; fi
    echo -n "[$arm] rep $rep " >> $OUT
    timeout -k 10 300 python3 tools/family_times.py $m 'variant=scan,wg_size=256,items_per_thread=8,tile_width=2048' 'variant=scan,wg_size=256,items_per_thread=8,tile_width=-1' 'variant=scan,wg_size=512,items_per_thread=4,tile_width=2048' 'variant=scan,wg_size=256,items_per_thread=4,tile_width=1024' 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%s %s: %.3f us' % (r['matrix'], r['spec'].replace('variant=scan,', ''), r['usec_cold']), end='   ')
print()" >> $OUT
  done
done
done
unset CASK_HIP_SCAN_PAD
for t in 2048; do timeout -k 10 200 python3 tools/stamps.py 256 8 webbase2 $t scan 2>/dev/null; done > gpurun_out/scan_stamps_pad.txt 2>&1
cat $OUT; head -12 gpurun_out/scan_stamps_pad.txt

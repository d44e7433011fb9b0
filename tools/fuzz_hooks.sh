#!/bin/bash
# The extractor sweep under each test hook (the overflow / fallback paths): tools/fuzz_hooks.sh SECONDS SEED
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SECS=${1:-100}; SEED=${2:-1}
OUT=gpurun_out/fuzz_campaign.txt
mkdir -p gpurun_out
for h in "ORBGPU_DEBUG_QT_KEYS=500" "ORBGPU_DEBUG_QT_KEYS=500 ORBGPU_DEBUG_QT_NOPRE=1" "ORBGPU_DEBUG_QT_NOPRE=1" "ORBGPU_DEBUG_QT_BATCH=1" "ORBGPU_DEBUG_FAST_QUEUE=16 ORBGPU_DEBUG_QT_KEYS=0" "ORBGPU_FAST_EARLY_OUT=1" "ORBGPU_FAST_EARLY_OUT=1 ORBGPU_DEBUG_FAST_QUEUE=16"; do
  line=$(env $h timeout -k 10 $((SECS + 120)) python tools/fuzz_extract.py $SECS $SEED 2>&1 | tail -1)
  echo "fuzz_extract [$h]: $line" | tee -a $OUT
  case "$line" in "fuzz ok"*) ;; *) echo "stopped: fuzz_extract under $h did not finish clean" | tee -a $OUT; exit 1;; esac
done
# the dense map's merge with the sampled long-range path forced on every tile / on most tiles
for h in "ORBGPU_DEBUG_MERGE_CAP=0" "ORBGPU_DEBUG_MERGE_CAP=700"; do
  line=$(env $h timeout -k 10 $((SECS + 120)) python tools/fuzz_cloud.py $SECS $SEED 2>&1 | tail -1)
  echo "fuzz_cloud [$h]: $line" | tee -a $OUT
  case "$line" in "fuzz ok"*) ;; *) echo "stopped: fuzz_cloud under $h did not finish clean" | tee -a $OUT; exit 1;; esac
done

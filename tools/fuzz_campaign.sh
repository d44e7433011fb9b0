#!/bin/bash
# Runs every randomised parity sweep (GPU path vs oracle) for SECONDS each and writes one summary with the build id:
#     tools/fuzz_campaign.sh SECONDS SEED [extra env, e.g. ORBGPU_DEBUG_QT_KEYS=500]
# Output: gpurun_out/fuzz_campaign.txt (copy to profiles/rNN_fuzz_campaign.txt).  Stops at the first failing sweep.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SECS=${1:-120}; SEED=${2:-1}; shift 2 || true
OUT=gpurun_out/fuzz_campaign.txt
mkdir -p gpurun_out
{
  echo "fuzz campaign: $SECS s per sweep, seed $SEED, env: $*"
  echo "liborbgpu.so sha256: $(sha256sum orb_slam2_map_amd/liborbgpu.so | cut -c1-16)  liborb_oracle.so sha256: $(sha256sum oracle/liborb_oracle.so | cut -c1-16)"
  echo "sources sha256 (csrc/*.hip, *.h): $(cat orb_slam2_map_amd/csrc/*.hip orb_slam2_map_amd/csrc/*.h | sha256sum | cut -c1-16)"
} >> $OUT
for s in ${FUZZ_SWEEPS:-fuzz_extract fuzz_projection fuzz_proj_variants fuzz_table fuzz_bf fuzz_bow fuzz_m6 fuzz_cloud}; do
  line=$(env "$@" timeout -k 10 $((SECS + 120)) python tools/$s.py $SECS $SEED 2>&1 | tail -1)
  echo "$s: $line" | tee -a $OUT
  case "$line" in "fuzz ok"*) ;; *) echo "campaign stopped: $s did not finish clean" | tee -a $OUT; exit 1;; esac
done

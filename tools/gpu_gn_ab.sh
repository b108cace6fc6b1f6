#!/bin/bash
# Same-box A/B of the single-launch GroupNorm threshold (T2V_GN_FUSED_SLICE, bytes of one statistics slice).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 65536 131072 262144 65536; do
  T2V_GN_FUSED_SLICE=$v timeout 300 python tools/profile_unet.py > gpurun_out/gn_ab_$v.log 2>&1
  echo "== slice $v"; sed -n 4p gpurun_out/gn_ab_$v.log; grep -E "^groupnorm" gpurun_out/gn_ab_$v.log
done

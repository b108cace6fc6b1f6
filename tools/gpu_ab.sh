#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.py): per-kind UNet step profile with each, interleaved.
#   gpurun -- 'bash tools/gpu_ab.sh old "" noprefetch nostrip old ""'     ("" = the default library)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
k=0
for tag in "$@"; do
  k=$((k+1))
  if [ -z "$tag" ]; then unset T2V_LIB_PATH; name=default; else export T2V_LIB_PATH=$GRAFT_REPO_ROOT/sd-webui-text2video_amd/libt2v_hip_$tag.so; name=$tag; fi
  timeout 300 python tools/profile_unet.py > gpurun_out/ab_${k}_$name.log 2>&1
  echo "== run $k: $name"; sed -n 4,5p gpurun_out/ab_${k}_$name.log; grep -E "^(gemm|groupnorm|attention|layernorm)" gpurun_out/ab_${k}_$name.log | awk '{printf "   %-14s %8s ms\n", $1, $2}'
  grep -E "\(0, 49152, 320, 320, 1, 0\)|\(0, 49152, 2560, 320, 1, 1\)|\(0, 12288, 5120, 640, 1, 1\)|\(2, 49152, 320, 960, 1, 0\)|\(1, 49152, 320, 2880, 1, 0\)" gpurun_out/ab_${k}_$name.log
done

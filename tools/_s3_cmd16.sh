export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0 1 0; do
  T2V_GN_STRIPS_R6=$v timeout 300 python tools/profile_unet.py 125 32 32 2 modelscope 2>&1 | grep -E "^geometry|^gemm|^groupnorm|^layernorm" | sed "s/^/[strips=$v] /" | cut -c1-160
done
for v in 1 0 1 0; do
  T2V_GN_STRIPS_R6=$v timeout 300 python tools/profile_unet.py 24 72 128 2 modelscope 2>&1 | grep -E "^geometry|^gemm|^groupnorm|^layernorm" | sed "s/^/[strips=$v] /" | cut -c1-160
done
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fullsize.py -rP -k "c2_125f_forward or c3_zeroscope_xl_forward or c3_zeroscope_xl_24" > gpurun_out/s3_strips_parity.log 2>&1; echo "parity exit $?"; grep -E "DEPLOYED|24 frames|passed|failed" gpurun_out/s3_strips_parity.log | cut -c1-220 | tail -n 8

export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for t in 0 300 600 1000 1500 0; do
  T2V_STAGGER_TICKS=$t timeout 300 python tools/profile_unet.py 24 32 32 2 modelscope > gpurun_out/s3_stagger_$t.log 2>&1
  echo "== stagger $t ticks"; grep -E "^geometry" gpurun_out/s3_stagger_$t.log; grep -E "gemm/plain +hbm +49152|\(0, 49152, 320, 320, 1, 0\)|\(0, 49152, 320, 640, 1, 0\)|\(0, 49152, 320, 1280" gpurun_out/s3_stagger_$t.log | cut -c1-120
done

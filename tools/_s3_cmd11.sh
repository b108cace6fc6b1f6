export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
# 1024x576 row counts (24 frames x 9216 / 2304 / 576 / 144 tokens, b = 2) on the sweep's square-frame shapes: 216 "frames" of 32x32
for lvl in L0 L1 L2 L3; do
  SWEEP_FRAMES=216 SWEEP_BATCH=2 timeout 900 python tools/gemm_sweep.py $lvl > gpurun_out/s3_sweep_b2_f216_$lvl.txt 2>&1
done
python tools/sweep_vs_policy.py gpurun_out/s3_sweep_b2_f216_L*.txt
# VideoCrafter rows (16 frames, b = 2) = the 32-frame b = 1 sweep; 48 / 32 frames b = 2 for the in-between clips
for fr in 48; do
  for lvl in L0 L1 L2 L3; do
    SWEEP_FRAMES=$fr SWEEP_BATCH=2 timeout 900 python tools/gemm_sweep.py $lvl > gpurun_out/s3_sweep_b2_f${fr}_$lvl.txt 2>&1
  done
  python tools/sweep_vs_policy.py gpurun_out/s3_sweep_b2_f${fr}_L*.txt
done

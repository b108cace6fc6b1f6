export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for fr in 24 12; do
  for lvl in L0 L1 L2 L3; do
    SWEEP_FRAMES=$fr SWEEP_BATCH=1 timeout 900 python tools/gemm_sweep.py $lvl > gpurun_out/s3_sweep_b1_f${fr}_$lvl.txt 2>&1
  done
  SWEEP_RANK=1 python tools/sweep_vs_policy.py gpurun_out/s3_sweep_b1_f${fr}_L*.txt
done
# VideoCrafter's own rows: 16 frames, b = 2
for lvl in L0 L1 L2 L3; do
  SWEEP_FRAMES=16 SWEEP_BATCH=2 timeout 900 python tools/gemm_sweep.py $lvl > gpurun_out/s3_sweep_b2_f16_$lvl.txt 2>&1
done
python tools/sweep_vs_policy.py gpurun_out/s3_sweep_b2_f16_L*.txt

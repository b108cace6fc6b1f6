export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "24 2" "125 2"; do
  set -- $cfg
  for lvl in L0 L1 L2 L3; do
    SWEEP_FRAMES=$1 SWEEP_BATCH=$2 timeout 900 python tools/gemm_sweep.py $lvl > gpurun_out/s3_sweep_b$2_f$1_$lvl.txt 2>&1
  done
  python tools/sweep_vs_policy.py gpurun_out/s3_sweep_b$2_f$1_L*.txt
done

export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fake_rccl.py -rP > gpurun_out/s3_fake_rccl_uncached.log 2>&1; echo "fake_rccl (uncached windows) exit $?"; grep -E "peer window|passed|failed|Error|error" gpurun_out/s3_fake_rccl_uncached.log | tail -n 6 | cut -c1-400
for fr in 6 32; do
  for lvl in L0 L1 L2 L3; do
    SWEEP_FRAMES=$fr SWEEP_BATCH=1 timeout 600 python tools/gemm_sweep.py $lvl > gpurun_out/s3_sweep_b1_f${fr}_$lvl.txt 2>&1
    cut -c1-420 gpurun_out/s3_sweep_b1_f${fr}_$lvl.txt
  done
done

export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_ops.py -k "reshard" > gpurun_out/s3_reshard.log 2>&1; echo "reshard exit $?"; tail -n 2 gpurun_out/s3_reshard.log
T2V_TEST_FULL=1 timeout 1500 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fake_rccl.py -rP > gpurun_out/s3_fake_rccl_full.log 2>&1; echo "fake_rccl full exit $?"; grep -E "library collectives|passed|failed|Error|error" gpurun_out/s3_fake_rccl_full.log | tail -n 12 | cut -c1-700
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_e2e.py -k "tsharded" > gpurun_out/s3_tshard_e2e.log 2>&1; echo "tsharded e2e exit $?"; tail -n 2 gpurun_out/s3_tshard_e2e.log
for g in "24 4 1" "125 4 1"; do
  timeout 300 python tools/profile_tshard_rank.py $g > "gpurun_out/s3_tshard_rank_$(echo $g | tr ' ' '_').log" 2>&1
  echo "== $g"; grep -A12 -E "T-shard rank" "gpurun_out/s3_tshard_rank_$(echo $g | tr ' ' '_').log" | cut -c1-300
done
T2V_REHEARSAL_ONLY=fake_rccl bash tools/gpu_rehearsal.sh
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fullsize.py tests/test_gpu_e2e.py tests/test_gpu_boundary.py -rP -k "vae or decode or infer or vid2vid" > gpurun_out/s3_vae_tests.log 2>&1; echo "vae tests exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/s3_vae_tests.log | cut -c1-220 | tail -n 14
for v in 96 0 96 0; do
  T2V_VAE_STRIPS_MB=$v timeout 300 python tools/profile_vae.py 24 32 32 2>&1 | grep -E "^VAE decode|^groupnorm|^gemm" | sed "s/^/[strips $v] /" | cut -c1-200
  T2V_VAE_STRIPS_MB=$v timeout 300 python tools/profile_vae.py 1 72 128 2>&1 | grep -E "^VAE decode|^groupnorm|^gemm" | sed "s/^/[strips $v] /" | cut -c1-200
done

export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_ops.py -k "groupnorm or shard or halo or gn" > gpurun_out/s3_gn_ops.log 2>&1; echo "gn ops exit $?"; tail -n 3 gpurun_out/s3_gn_ops.log
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fake_rccl.py tests/test_gpu_e2e.py tests/test_gpu_multiproc.py -rP -k "tsharded or library_collectives or runner_layouts" > gpurun_out/s3_gn_shard.log 2>&1; echo "sharded exit $?"; grep -E "passed|failed|Error" gpurun_out/s3_gn_shard.log | tail -n 4 | cut -c1-300
for g in "24 4 1" "125 4 1"; do
  for v in 1 0; do
    T2V_GN_PHASE1_TICKET=$v timeout 300 python tools/profile_tshard_rank.py $g 2>&1 | grep -E -A3 "T-shard rank" | cut -c1-330 | sed "s/^/[ticket=$v] /"
  done
done

export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for g in "24 32 32 1" "12 32 32 1" "6 32 32 1"; do
  for v in 1 0; do
    T2V_TILE_R6=$v timeout 300 python tools/profile_unet.py $g modelscope 2>&1 | grep -E "^geometry" | sed "s/^/[r6=$v] /"
  done
done
for g in "24 4 1" "125 4 1" "24 2 0"; do
  for v in 1 0; do
    T2V_TILE_R6=$v timeout 300 python tools/profile_tshard_rank.py $g 2>&1 | grep -E "T-shard rank" | cut -c1-330 | sed "s/^/[r6=$v] /"
  done
done
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_e2e.py tests/test_gpu_multiproc.py -k "tsharded or runner_layouts or pair" > gpurun_out/s3_final_shard.log 2>&1; echo "sharded tests exit $?"; tail -n 2 gpurun_out/s3_final_shard.log

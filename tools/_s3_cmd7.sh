export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_videocrafter.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -rP -k "c4 or videocrafter or lvdm or tsharded" > gpurun_out/s3_r6tiles_tests.log 2>&1; echo "tests exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/s3_r6tiles_tests.log | cut -c1-200 | tail -n 12
for v in 1 0 1 0; do
  T2V_TILE_R6=$v timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lvdm bench r6=$v', d['value'], d['roofline']['unet_step_ms'], d['roofline']['unet_step_ms_events'])"
done
for g in "24 4 1" "125 4 1" "24 2 0"; do
  for v in 1 0; do
    T2V_TILE_R6=$v timeout 300 python tools/profile_tshard_rank.py $g 2>&1 | grep -E "T-shard rank" | cut -c1-330 | sed "s/^/[r6=$v] /"
  done
done
for g in "6 32 32 1" "12 32 32 1"; do
  for v in 1 0; do
    T2V_TILE_R6=$v timeout 300 python tools/profile_unet.py $g modelscope 2>&1 | grep -E "^geometry" | sed "s/^/[r6=$v] /"
  done
done

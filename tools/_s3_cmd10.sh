export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0 1 0; do
  T2V_TILE_R6=$v timeout 300 python tools/profile_unet.py 24 32 32 2 modelscope 2>&1 | grep -E "^geometry" | sed "s/^/[r6=$v] /"
done
for v in 1 0; do
  T2V_TILE_R6=$v timeout 300 python tools/profile_unet.py 125 32 32 2 modelscope 2>&1 | grep -E "^geometry|^gemm|^groupnorm|^layernorm|^attention" | sed "s/^/[r6=$v] /"
  T2V_TILE_R6=$v timeout 300 python tools/profile_unet.py 24 72 128 2 modelscope 2>&1 | grep -E "^geometry|^gemm|^groupnorm|^layernorm|^attention" | sed "s/^/[r6=$v] /"
done
for v in 1 0; do
  T2V_TILE_R6=$v timeout 600 python bench.py --frames 125 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('125f bench r6=$v', d['value'], d['roofline']['unet_step_ms'], d['roofline']['frac'])"
  T2V_TILE_R6=$v timeout 600 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('XL bench r6=$v', d['value'], d['roofline']['unet_step_ms'], d['roofline']['frac'])"
done
timeout 1500 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fullsize.py -rP -k "c2_125f_forward or c3_zeroscope_xl_forward or c1_24f_forward or c3_zeroscope_xl_24" > gpurun_out/s3_r6big_parity.log 2>&1; echo "parity exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/s3_r6big_parity.log | cut -c1-220 | tail -n 10

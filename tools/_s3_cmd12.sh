export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0; do
  T2V_TILE_R6=$v timeout 600 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('XL bench r6=$v', d['value'], d['roofline']['unet_step_ms'], d['roofline']['frac'])"
done
timeout 600 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fullsize.py -rP -k "c3_zeroscope_xl" > gpurun_out/s3_xl_parity.log 2>&1; echo "xl parity exit $?"; grep -E "rel-L2|passed|failed" gpurun_out/s3_xl_parity.log | cut -c1-220 | tail -n 8
bash tools/gpu_pass.sh suite

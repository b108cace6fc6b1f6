for cfg in "T2V_PRECISE_ATTN=1 T2V_PRECISE_RESAMPLE=0" "T2V_PRECISE_ATTN=0 T2V_PRECISE_RESAMPLE=1" "T2V_PRECISE_ATTN=0 T2V_PRECISE_RESAMPLE=0"; do
  echo "== $cfg"
  env $cfg timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_fullsize.py -rP -k "c4" 2>&1 | grep -E "DEPLOYED" | cut -c1-200
  env $cfg timeout 400 python bench.py --model lvdm --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   bench', d['value'], d['roofline']['unet_step_ms_events'])"
done

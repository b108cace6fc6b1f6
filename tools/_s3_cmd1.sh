export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_ops.py -k "attention" > gpurun_out/s3_attn.log 2>&1; echo "attn tests exit $?"; tail -n 3 gpurun_out/s3_attn.log
for v in 1 0; do
  T2V_ATTN2=$v timeout 600 python bench.py --height 576 --width 1024 --steps 1 --warmup 1 --no-cpu-baseline --also-batched 0 > gpurun_out/s3_bench_xl_attn2_$v.json 2> gpurun_out/s3_bench_xl_attn2_$v.err; echo "bench XL attn2=$v exit $?"; cut -c1-200 gpurun_out/s3_bench_xl_attn2_$v.json
done
for v in 1 0 1 0; do
  T2V_ATTN2=$v timeout 300 python tools/profile_unet.py 24 32 32 2 modelscope > gpurun_out/s3_prof_attn2_$v.log 2>&1; sed -n 4,5p gpurun_out/s3_prof_attn2_$v.log; grep -E "^attention" gpurun_out/s3_prof_attn2_$v.log
done
bash tools/gpu_pass.sh suite

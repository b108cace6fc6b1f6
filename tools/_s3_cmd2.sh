export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
./tools/ubench/cumask_probe > gpurun_out/s3_cumask.txt 2>&1; cat gpurun_out/s3_cumask.txt
timeout 900 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fake_rccl.py -rP > gpurun_out/s3_fake_rccl.log 2>&1; echo "fake_rccl exit $?"; grep -E "library collectives|passed|failed|Error|error" gpurun_out/s3_fake_rccl.log | tail -n 12 | cut -c1-900
for m in "b2" "b1" "streams" "mask halves" "mask interleaved" "mask xcdsplit" "b2"; do
  T2V_GN_EPI=0 timeout 300 python tools/two_stream_probe.py $m 2>&1 | grep -E "ms per guided|Error|error|assert" | sed 's/^/[gn_epi=0] /'
done
for m in "b2" "mask halves" "mask interleaved" "mask xcdsplit" "b2"; do
  timeout 300 python tools/two_stream_probe.py $m 2>&1 | grep -E "ms per guided|Error|error|assert" | sed 's/^/[fused]    /'
done

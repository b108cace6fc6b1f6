export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest -q --tb=short -p no:cacheprovider tests/test_gpu_fake_rccl.py -rP -k "bounded" > gpurun_out/s3_peer_fault.log 2>&1; echo "peer fault exit $?"; grep -E "peer window|passed|failed|Error|error" gpurun_out/s3_peer_fault.log | tail -n 12 | cut -c1-900
for m in "b2" "mask halves" "streams"; do
  T2V_GN_EPI=0 T2V_GN_COOP=0 timeout 300 python tools/two_stream_probe.py $m 2>&1 | grep -E "ms per guided|Error|error|assert" | sed 's/^/[gn_epi=0 coop=0] /'
done

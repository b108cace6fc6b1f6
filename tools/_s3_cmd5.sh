export T2V_EXPERIMENTAL=1 TMPDIR=/tmp
mkdir -p gpurun_out
for g in "24 4" "125 4"; do
  timeout 900 python tools/tshard_e2e_wall.py $g 2>&1 | grep -E "frames over|Error|error|Traceback" | cut -c1-600
  for idx in 0 1 3; do
    T2V_GN_COOP=0 T2V_GN_EPI=0 timeout 300 python tools/profile_tshard_rank.py $g $idx 2>&1 | grep -E "T-shard rank" | cut -c1-330 | sed 's/^/[compute only, norms unfused] /'
  done
done

#!/bin/bash
# HBM traffic counters (separate passes, MI355X_MICROARCH.md §HBM)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/tools/pmc_target.py 2 > $R/gpurun_out/pmc_$C.log 2>&1
  echo "$C exit $?"; tail -n 2 $R/gpurun_out/pmc_$C.log
done
cd $R
python tools/pmc_post.py gpurun_out > gpurun_out/pmc_traffic.txt 2>&1; head -n 30 gpurun_out/pmc_traffic.txt
# keep only the summaries (raw CSVs can be large)
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete

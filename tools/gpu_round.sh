#!/bin/bash
# One GPU-box session: parity tests, smoke, per-op profile.  Logs go to gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || echo "BUILD FAILED"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"
tail -n 60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 5 gpurun_out/smoke.log
timeout 600 python tools/profile_unet.py 24 32 32 2 > gpurun_out/profile_unet.log 2>&1; echo "profile exit $?"; tail -n 75 gpurun_out/profile_unet.log

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/determinism.py > gpurun_out/determinism_tiny.log 2>&1; echo "determinism exit $?"; tail -n 40 gpurun_out/determinism_tiny.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 5 gpurun_out/smoke.log

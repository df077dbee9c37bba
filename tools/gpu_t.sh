#!/bin/bash
# usage: gpu_t.sh TAG "pytest args"   -- a subset of the GPU tests, log under gpurun_out/TAG/
TAG=$1; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest $@ -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  |decoder parity" gpurun_out/$TAG/pytest.log | cut -c1-400 | head -40

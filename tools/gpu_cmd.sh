#!/bin/bash
# generic: TAG then a command line; stdout+stderr to gpurun_out/TAG/out.log
TAG=$1; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 800 "$@" > gpurun_out/$TAG/out.log 2>&1; echo "exit $?"; grep -v amdgpu.ids gpurun_out/$TAG/out.log | tail -60 | cut -c1-600

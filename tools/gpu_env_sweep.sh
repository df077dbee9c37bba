#!/bin/bash
# does any HIP runtime switch change the replayed step?  (bench.py --steps 20 --no-extras, ms per step)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { ms=$(env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"); echo "$* -> $ms"; }
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run AMD_OPT_FLUSH=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run GPU_MAX_HW_QUEUES=1
run X=1

#!/bin/bash
# review r05 item 1: the per-node boundary, C micro-benchmark + the library's kernels / the whole step three ways
#   gpurun --timeout 900 -- 'bash tools/gpu_boundary.sh r06_boundary'
TAG=${1:-r06_boundary}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
B=$PWD/tools/ubench/boundary
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B tools/ubench/boundary.hip
timeout 120 $B 200 20 > $OUT/c_default.txt 2>&1
if [ "$2" = "quick" ]; then timeout 600 python tools/boundary_probe.py > $OUT/py_probe.txt 2>&1; tail -n 30 $OUT/py_probe.txt; exit 0; fi
HIP_FORCE_DEV_KERNARG=0 timeout 120 $B 200 20 > $OUT/c_dev_kernarg0.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 120 $B 200 20 > $OUT/c_dev_kernarg1.txt 2>&1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 120 $B 200 20 > $OUT/c_packet_capture0.txt 2>&1
AMD_OPT_FLUSH=0 timeout 120 $B 200 20 > $OUT/c_opt_flush0.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c -o c -- $B 200 3 > $OUT/c_under_rocprof.txt 2>&1)
find $OUT/prof_c -name '*kernel_stats*' | head -1 | xargs -r cat > $OUT/c_rocprof_kernel_stats.csv
rm -rf $OUT/prof_c
timeout 600 python tools/boundary_probe.py > $OUT/py_probe.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 600 python tools/boundary_probe.py > $OUT/py_probe_dev_kernarg1.txt 2>&1
tail -n 30 $OUT/c_default.txt $OUT/py_probe.txt

export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; mkdir -p gpurun_out/st2
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_st2 -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --overlap on > $R/gpurun_out/st2/rocprof.log 2>&1; echo "rocprof exit $?")
DB=$(ls /tmp/prof_st2/*.db /tmp/prof_st2/*/*.db 2>/dev/null | head -1)
python tools/graph_gaps.py $DB > gpurun_out/st2/graph_gaps.txt 2>&1
head -6 gpurun_out/st2/graph_gaps.txt
awk '/sequence of the last step/{f=1;next} f {print}' gpurun_out/st2/graph_gaps.txt | awk '{ if (NR>1) { gap = $2 - pe; if (gap > 3) print "GAP", gap, "before", $0 } pe = $2 + $3 }' | cut -c1-160 | head -20

export TMPDIR=/tmp
R=$PWD
for mode in "slab save" "slab"; do
  (cd /tmp && rm -rf /tmp/pf && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o t -- python $R/tools/ffn3_trace.py $mode > /tmp/pf.log 2>&1)
  DB=$(ls /tmp/pf/*.db /tmp/pf/*/*.db 2>/dev/null | head -1)
  echo "mode: $mode"; python $R/tools/prof_summary.py $DB 1 2>/dev/null | grep -E "ffn3_fwd" | cut -c1-140
  grep -E "total cycles|100 MHz" /tmp/pf.log | cut -c1-200
done

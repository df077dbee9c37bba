export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OTR_SWITCHES=ops._DEBUG_WQ=1 timeout 300 python bench.py --model conformer --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "^wq" | sort | uniq -c | sort -k2,3 

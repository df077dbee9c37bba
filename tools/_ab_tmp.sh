export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "layernorm or conformer or ln or c4 or norm" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --model conformer --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conformer', round(d['value'],1), round(d['ms_per_step'],3))"; done
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', round(d['value'],1), round(d['ms_per_step'],3))"; done

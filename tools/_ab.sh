cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k 'sets or golden or extractor' 2>&1 | tail -3
python bench.py --workload c3 --path sets --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sets c3', round(d['value']), round(d['ms_per_step'],2), d['stage_ms'])"
python bench.py --workload c2 --path sets --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sets c2', round(d['value']), round(d['ms_per_step'],2), d['stage_ms'])"

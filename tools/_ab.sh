cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 e2e', d['end_to_end'], 'serial', d['ms_per_step_one_at_a_time'])"
python bench.py --workload c2 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 e2e', d['end_to_end'], 'serial', d['ms_per_step_one_at_a_time'])"

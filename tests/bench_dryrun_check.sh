#!/bin/bash
# GPU box with ONE GPU: the multi-rank control flow of bench.py (rank environment, per-rank step size, balanced
# shards, barriers, max-over-ranks timing, rank-0 JSON) with two ranks sharing the GPU and the exchange step
# replaced by a count reduction (EG3D_BENCH_DRYRUN_GATHER=1). Checks that every step's cloud is the whole batch.
set -e
cd "$(dirname "$0")/.."
out=$(EG3D_BENCH_DRYRUN_GATHER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 2 --workload c2 --steps 6 --warmup 2 2>gpurun_out/dryrun.err | tail -1)
one=$(python bench.py --workload c2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
python - "$out" "$one" <<'PY'
import json, sys
two, one = json.loads(sys.argv[1]), json.loads(sys.argv[2])
assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "DRY RUN" in two["data"], two
assert two["config"]["edge_points_per_step"] == one["config"]["edge_points_per_step"], (two["config"], one["config"])
print("BENCH-DRYRUN-OK", two["config"]["edge_points_per_step"], "points per step on 2 ranks and on 1")
PY

#!/bin/bash
# usage: tests/bench_dryrun_check.sh [ranks, default 2; 8 = what the driver's scaling run launches]
# GPU box with ONE GPU: the multi-rank control flow of bench.py (rank environment, per-rank step size, StepPlan's balanced
# shards, barriers, max-over-ranks timing, rank-0 JSON) with N ranks sharing the GPU and the exchange step replaced by
# a count reduction (EG3D_BENCH_DRYRUN_GATHER=1). Checks that every step's cloud is the whole batch.
set -e
cd "$(dirname "$0")/.."
n=${1:-2}
mkdir -p gpurun_out
# (one step in flight per rank: N ranks x 4 contexts each would only multiply the work buffers on the shared GPU)
out=$(EG3D_BENCH_DRYRUN_GATHER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus $n --workload c2 --steps 6 --warmup 2 --batch-seeds 2000 --inflight 2 2>gpurun_out/dryrun_$n.err | tail -1)
one=$(python bench.py --workload c2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
python - "$out" "$one" "$n" <<'PY'
import json, sys
two, one, n = json.loads(sys.argv[1]), json.loads(sys.argv[2]), int(sys.argv[3])
assert two["n_gpus"] == n and two["scaling"] == "strong" and "DRY RUN" in two["data"], two
assert two["config"]["edge_points_per_step"] == one["config"]["edge_points_per_step"], (two["config"], one["config"])
print("BENCH-DRYRUN-OK", two["config"]["edge_points_per_step"], "points per step on %d ranks and on 1" % n)
PY

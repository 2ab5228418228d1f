#!/bin/bash
O=gpurun_out/r4_waves; mkdir -p $O; export TMPDIR=/tmp
for w in 3 4; do for wl in c3 c2; do
  EG3D_K3B_WAVES=$w timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-sublines > $O/${wl}_w$w.json 2> $O/${wl}_w$w.err
  python - $O/${wl}_w$w.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], "ms/step %.2f value %.4g serial %.2f k3b %.2f" % (d["ms_per_step"], d["value"], d.get("ms_per_step_one_at_a_time",0), d["roofline"]["kernel_ms_per_step"]))
P
done; done

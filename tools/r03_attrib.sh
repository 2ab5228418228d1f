#!/bin/bash
# usage (GPU box, repo root): tools/r03_attrib.sh "<variant names ('default' = libeg3d.so)>" [workload=c3]
# Per variant: one-step-at-a-time stage times and separate --pmc passes (FETCH_SIZE; WRITE_SIZE; TCC hit/miss;
# SQ activity incl. the VALU active-lane counters) of a short one-step-at-a-time run -> gpurun_out/attrib_<variant>.txt
wl=${2:-c3}
mkdir -p gpurun_out
for v in $1; do
  if [ "$v" = default ]; then unset EG3D_LIB; lib=""; else lib=edgegraph3d_amd/variants/libeg3d_$v.so; export EG3D_LIB=$PWD/$lib; fi
  o=gpurun_out/attrib_${v}_$wl.txt; : > $o
  args="--workload $wl --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"
  [ $wl = c4 ] && args="--workload c4 --batch-seeds ${C4_SEEDS:-2048} --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"
  python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],2), d['stage_ms'])" | tee -a $o
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
     "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" \
     "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" ${EXTRA_SETS:+"$EXTRA_SETS"}; do
    tools/pmc_pass.sh ${v}_x "$args" $set 2>&1 | grep -E "k3b|k3a|k4|rc=" | tee -a $o
  done
done

#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <workload: c3|c2|c4|c3real> <tag, e.g. r05>
# rocprofv3 passes of bench.py for one workload:
#   kt      --kernel-trace --stats of the driver's command for that workload (steps in flight; the C4 / C5 sub-lines of the
#           default line are left out with --no-sublines: they launch the same kernels on other workloads and would blur the
#           per-kernel averages)
#   serial  --kernel-trace --stats with ONE step on the GPU at a time (--inflight 1 --no-extras): the per-kernel average
#           that bench.py's roofline.kernel_ms_per_step (measured the same way, by HIP events) must agree with
#   fetch / write / sq / sq2   SEPARATE --pmc passes of the serial command, as the profiling guide prescribes
# Summaries -> gpurun_out/<tag>_<workload>_*; copy to profiles/ to commit.
wl=${1:-c3}; tag=${2:-r06}_$wl
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
[ -f $out/pmc_traffic.json ] || cp profiles/pmc_traffic.json $out/pmc_traffic.json  # the other workloads' entries are kept
# c4: a context's FIRST launch on a many-view scene is followed by a relaunch of the chains that outgrew their slices (round 6:
# only those); the per-launch means below are meant to describe the steady-state launch, so the profiled contexts start with
# the capacities every C4 context settles at (test knobs; the kernel and its launch are the steady-state ones)
[ "$wl" = c4 ] && export EG3D_CHAIN_CAP0=768 EG3D_POOL_CAP0=196608
case $wl in
  c4) kt_args="--workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"; pmc_args="--workload c4 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"; pmc_steps=3;;
  *)  kt_args="--workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-sublines"; pmc_args="--workload $wl --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"; pmc_steps=4;;
esac
run() { # name, bench args, rocprof flags...
  local name=$1; local args=$2; shift 2
  rm -rf $out/prof_$name
  timeout -k 5 900 rocprofv3 "$@" -d $out/prof_$name -o $tag -- python bench.py $args > $out/prof_$name.out 2> $out/prof_$name.err
  echo "pass $name rc=$?"
}
run kt "$kt_args" --kernel-trace --stats
run serial "$pmc_args" --kernel-trace --stats
run fetch "$pmc_args" --kernel-trace --pmc FETCH_SIZE
run write "$pmc_args" --kernel-trace --pmc WRITE_SIZE
run sq "$pmc_args" --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
run sq2 "$pmc_args" --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64
db() { find $out/prof_$1 -name "*_results.db" | head -1; }
python tools/profile_summary.py --tag $tag --out $out --workload $wl --pmc-steps $pmc_steps --pmc-cmd "python bench.py $pmc_args" \
  --kt "$(db kt)" --fetch "$(db fetch)" --write "$(db write)" --sq "$(db sq)" "$(db sq2)" --cmd "python bench.py $kt_args" | tail -40
python tools/profile_summary.py --tag ${tag}_serial --out $out --kt "$(db serial)" --cmd "python bench.py $pmc_args" | tail -12
tail -1 $out/prof_kt.out > $out/${tag}_bench_line.json
rm -rf $out/prof_kt $out/prof_serial $out/prof_fetch $out/prof_write $out/prof_sq $out/prof_sq2 $out/tmp/*

#!/bin/bash
# usage (GPU box, repo root): tools/profile_r03.sh <workload: c3|c2|c4|c3real|c2_sets|c3_sets> [tag=r03]
# (round 3: the C4 counter passes run at the bench's own step size, 8192 seeds, so that roofline.traffic belongs to the
# bench line; SQ_THREAD_CYCLES_VALU added: active-lane fraction = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))
# One rocprofv3 --kernel-trace --stats pass of the bench command the driver runs for that workload, and
# SEPARATE --pmc passes (FETCH_SIZE; WRITE_SIZE; two SQ sets) of a short one-step-at-a-time run, as the
# profiling guide prescribes. Summaries -> gpurun_out/<tag>_<workload>_*; copy to profiles/ to commit.
wl=${1:-c3}; tag=${2:-r03}_$wl
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
[ -f $out/pmc_traffic.json ] || cp profiles/pmc_traffic.json $out/pmc_traffic.json  # the other workloads' entries are kept
case $wl in
  *_sets) b=${wl%_sets}; kt_args="--workload $b --path sets --steps 6 --warmup 2 --no-cpu-baseline"; pmc_args="--workload $b --path sets --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"; pmc_steps=4;;
  c4) kt_args="--workload c4 --steps 3 --warmup 1 --no-cpu-baseline"; pmc_args="--workload c4 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"; pmc_steps=3;;
  *)  kt_args="--workload $wl --steps 20 --warmup 5 --no-cpu-baseline"; pmc_args="--workload $wl --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras"; pmc_steps=4;;
esac
run() { # name, bench args, rocprof flags...
  local name=$1; local args=$2; shift 2
  rm -rf $out/prof_$name
  timeout -k 5 900 rocprofv3 "$@" -d $out/prof_$name -o $tag -- python bench.py $args > $out/prof_$name.out 2> $out/prof_$name.err
  echo "pass $name rc=$?"
}
run kt "$kt_args" --kernel-trace --stats
run fetch "$pmc_args" --kernel-trace --pmc FETCH_SIZE
run write "$pmc_args" --kernel-trace --pmc WRITE_SIZE
run sq "$pmc_args" --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
run sq2 "$pmc_args" --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64
find $out/prof_kt -name "*.db" | head -1 > $out/.kt_db
db() { find $out/prof_$1 -name "*_results.db" | head -1; }
python tools/profile_summary.py --tag $tag --out $out --workload $wl --pmc-steps $pmc_steps --pmc-cmd "python bench.py $pmc_args" \
  --kt "$(db kt)" --fetch "$(db fetch)" --write "$(db write)" --sq "$(db sq)" "$(db sq2)" --cmd "python bench.py $kt_args" | tail -40
tail -1 $out/prof_kt.out > $out/${tag}_bench_line.json
rm -rf $out/prof_kt $out/prof_fetch $out/prof_write $out/prof_sq $out/prof_sq2 $out/tmp/*

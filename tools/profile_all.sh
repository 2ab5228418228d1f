#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_all.sh <tag> [bench args...]
# Runs the rocprofv3 passes the bench numbers are checked against and writes the summaries to
# gpurun_out/<tag>_*: one --kernel-trace --stats pass, and SEPARATE --pmc passes (FETCH_SIZE,
# WRITE_SIZE, two SQ sets) each with --kernel-trace only, as the profiling guide prescribes.
tag=${1:-r01}; shift
args=${@:---steps 5 --warmup 1 --no-cpu-baseline}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=$out/tmp; mkdir -p $TMPDIR
run() { # name, rocprof flags...
  local name=$1; shift
  rm -rf $out/prof_$name
  timeout -k 5 600 rocprofv3 "$@" -d $out/prof_$name -o $tag -- python bench.py $args > $out/prof_$name.out 2> $out/prof_$name.err
  echo "pass $name rc=$?"
}
run kt --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python tools/profile_summary.py --tag $tag --out $out --kt $out/prof_kt/${tag}_results.db --fetch $out/prof_fetch/${tag}_results.db \
  --write $out/prof_write/${tag}_results.db --sq $out/prof_sq/${tag}_results.db $out/prof_sq2/${tag}_results.db --cmd "python bench.py $args"

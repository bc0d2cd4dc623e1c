#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh OUTDIR [bench args]   (prev = tools/librmhmc_hip_prev.so via RMHMC_HIP_LIB)
O=$1; shift; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0 $@"
for rep in 1 2; do
  RMHMC_HIP_LIB=$PWD/tools/librmhmc_hip_prev.so timeout -k 10 300 $B > $O/prev$rep.json 2> $O/prev$rep.err || { echo "prev failed"; tail -3 $O/prev$rep.err; exit 1; }
  timeout -k 10 300 $B > $O/new$rep.json 2> $O/new$rep.err || { echo "new failed"; tail -3 $O/new$rep.err; exit 1; }
done
python tools/summ_bench.py $O/prev1.json $O/new1.json $O/prev2.json $O/new2.json

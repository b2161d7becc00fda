#!/bin/bash
# tools/ab_trees.sh TAG ROUNDS TREE... : same-box alternating A/B of whole source trees (each with its own package + library, e.g.
# `git archive COMMIT | tar -x -C build/wt/COMMIT` + make): the headline's driver-form value without companions; "." = this tree
T=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$T
out=$R/gpurun_out/$T/ab_trees.txt
: > $out
for r in $(seq 1 $N); do
  for t in "$@"; do
    v=$(cd $R/$t && timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-companions --no-h2d 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "round $r tree $t $v" | tee -a $out
  done
done

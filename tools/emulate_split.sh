#!/bin/bash
# Compute-side strong-scaling ceiling of the framebuffer split, on ONE GPU: render each rank's share of an N-way split alone
# (bench.py --emulate-split r/N, no collective) and compare the slowest share with the whole frame.
# Usage: tools/emulate_split.sh <N> <out.txt> <bench args...>
N=$1; OUT=$2; shift 2; mkdir -p $(dirname $OUT)
COMMON="--no-cpu-baseline --no-pmc --no-parity"
t1=$(python bench.py $COMMON "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
worst=0; all=""
for r in $(seq 0 $((N-1))); do
  t=$(python bench.py $COMMON "$@" --emulate-split $r/$N 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  all="$all $t"
  worst=$(python -c "print(max($worst, $t))")
done
echo "[$*] N=$N whole $t1 ms/step; shares:$all; worst $worst -> ceiling $(python -c "print(round($t1/$worst,2))")x" | tee -a $OUT

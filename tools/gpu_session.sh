#!/bin/bash
# One GPU-box session: the -m gpu tests, then a list of bench.py runs (one JSON line each).  A step that times out ends the session
# (no further GPU step after a hang).  Usage: tools/gpu_session.sh <tag> [--no-tests] -- "<bench args>" "<bench args>" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
RUN_TESTS=1
if [ "$1" == "--no-tests" ]; then RUN_TESTS=0; shift; fi
if [ "$1" == "--" ]; then shift; fi
if [ $RUN_TESTS == 1 ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
  tail -3 $OUT/tests.log
  if [ $rc == 124 ] || [ $rc == 137 ]; then echo "tests timed out: stopping"; exit 1; fi
fi
i=0
for args in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py $args > $OUT/bench$i.json 2> $OUT/bench$i.err; rc=$?
  echo "bench$i [$args] rc=$rc: $(python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench$i.json").read().strip().splitlines()[-1])
    print(d["value"], "Mrays/s", d["config"]["ms_per_frame"], "ms/frame", {k: v["ms"] for k, v in d["kernels"].items()})
except Exception as e:
    print("no json:", e)
PY
)"
  if [ $rc == 124 ] || [ $rc == 137 ]; then echo "bench timed out: stopping"; exit 1; fi
done

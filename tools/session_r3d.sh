#!/bin/bash
mkdir -p gpurun_out/r3d
python3 -c "
from simple_raytracer_amd import lib
for it in (500, 2000, 4000): print('valu_rate', it, lib.valu_rate(it))" > gpurun_out/r3d/valu_rate.txt 2>&1
cat gpurun_out/r3d/valu_rate.txt
bash tools/gpu_session.sh r3d -- \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 42" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 44" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 45" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 41" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --variant 44" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --variant 44 --lights 4" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --variant 42 --lights 4" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload cube_ground --variant 42" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload cube_ground --variant 44"

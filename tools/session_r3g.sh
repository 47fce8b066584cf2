#!/bin/bash
bash tools/gpu_session.sh r3g --no-tests -- \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 45" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 45" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --variant 45" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --streams 1" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --streams 1 --variant 45" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload cube_ground --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload cube_ground --streams 1 --variant 45"

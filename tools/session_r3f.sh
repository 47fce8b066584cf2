#!/bin/bash
bash tools/gpu_session.sh r3f -- \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --streams 1" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --streams 1 --variant 44" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --lights 4" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --lights 16" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload cube_ground" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload main_nocats --width 3840 --height 2160 --lights 64 --frames 8 --steps 5"

#!/bin/bash
bash tools/gpu_session.sh r3e -- \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --variant 43" \
  "--workload soup --frames 4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline" \
  "--workload soup --width 2048 --height 2048 --frames 4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-parity" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --lights 4" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --lights 16" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --workload main_nocats --width 3840 --height 2160 --lights 64 --frames 8 --steps 5"

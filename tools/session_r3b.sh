#!/bin/bash
# round-3 session b: node-major A/B + vector-memory counters of k_trace_nq (wide vs narrow)
bash tools/gpu_session.sh r3b -- \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 40" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 41" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 --variant 42" \
  "--no-pmc --no-cpu-baseline --no-soup --no-parity --variant 42" \
  "--workload soup --frames 4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-parity --variant 41" \
  "--workload soup --frames 4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-parity --variant 42" \
  "--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --variant 42" || exit 1
for v in 0 40 42; do
  PMC_GROUPS=v PMC_STEPS=3 timeout -k 10 400 bash profiles/pmc_run.sh gpurun_out/r3b/pmc_v$v --streams 1 --frames 8 --variant $v > gpurun_out/r3b/pmc_v$v.log 2>&1
  python3 profiles/pmc_summary.py gpurun_out/r3b/pmc_v$v > gpurun_out/r3b/pmc_v$v.txt 2>&1
  rm -rf gpurun_out/r3b/pmc_v$v
done

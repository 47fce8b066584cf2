#!/bin/bash
mkdir -p gpurun_out/r3k
nproc; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
python3 tools/build_throughput_probe.py 2>&1 | tee gpurun_out/r3k/build_probe.txt
timeout -k 10 400 python examples/frame_pipeline.py --builders 4 --serial-builders 6,8,12 > gpurun_out/r3k/frame_pipeline.txt 2>&1; tail -8 gpurun_out/r3k/frame_pipeline.txt

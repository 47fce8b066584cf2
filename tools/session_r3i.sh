#!/bin/bash
mkdir -p gpurun_out/r3i
timeout -k 10 300 python examples/frame_pipeline.py --builders 4,6,8 > gpurun_out/r3i/frame_pipeline.txt 2>&1; tail -12 gpurun_out/r3i/frame_pipeline.txt
bash tools/collect_profiles_r3.sh r03 2>&1 | tail -5
bash tools/emulate_split.sh 8 gpurun_out/r3i/ceil.txt --workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5
bash tools/emulate_split.sh 8 gpurun_out/r3i/ceil.txt --no-soup

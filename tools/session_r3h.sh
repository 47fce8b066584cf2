#!/bin/bash
mkdir -p gpurun_out/r3h
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/tests.log 2>&1; rc=$?; tail -15 gpurun_out/r3h/tests.log
if [ $rc != 0 ]; then exit 1; fi
timeout -k 10 300 python examples/frame_pipeline.py > gpurun_out/r3h/frame_pipeline.txt 2>&1; tail -25 gpurun_out/r3h/frame_pipeline.txt

#!/bin/bash
# Collects, on the GPU box, what profiles/ keeps per round: rocprofv3 --kernel-trace --stats and the --pmc passes (profiles/pmc_run.sh,
# one counter group per pass) of bench.py for the three heavy configurations.  Usage: tools/collect_profiles.sh <tag>
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}_profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A CFG
CFG[k3]="--no-soup"
CFG[k4]="--workload k4 --width 3840 --height 2160 --lights 64 --frames 12"
CFG[k5]="--workload soup --width 2048 --height 2048 --frames 4"
CFG[k5_1080p]="--workload soup --frames 4"
for k in k3 k4 k5 k5_1080p; do
  # kernels ALONE (one stream): the durations the roofline block quotes; then the default command (frames overlapped on 4 streams)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${k}_stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --streams 1 ${CFG[$k]} > $OUT/${k}_s1_bench.json 2> $OUT/${k}_stats.log || echo "stats $k failed"
  cp $(ls $OUT/${k}_stats/*/*kernel_stats.csv | head -1) $OUT/${k}_kernel_stats.csv 2>/dev/null
  python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc ${CFG[$k]} > $OUT/${k}_bench.json 2> $OUT/${k}_bench.log || echo "bench $k failed"
done
for k in k3 k4 k5; do
  PMC_GROUPS=s bash $ROOT/profiles/pmc_run.sh gpurun_out/${TAG}_profiles/${k}_pmc ${CFG[$k]} --streams 1 --steps 4 --warmup 1 > $OUT/${k}_pmc.log 2>&1
done
echo collected

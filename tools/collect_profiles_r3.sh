#!/bin/bash
# Round 3: what profiles/ keeps.  Per configuration: rocprofv3 --kernel-trace --stats of bench.py with --streams 1 (kernels alone: the
# durations the roofline block quotes), the default bench line (frames overlapped on 4 streams), and counter passes of the lean probe
# (tools/pmc_probe.py: the same scene and kernels through srt_render, one counter group per pass).  Usage: tools/collect_profiles_r3.sh [tag]
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}_profiles
mkdir -p $OUT
(cd $ROOT && python3 -m simple_raytracer_amd.build > /dev/null && python3 -c "from oracle import pyoracle; pyoracle.build()" > /dev/null) || { echo "build failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
declare -A CFG PROBE
CFG[k3]="--no-soup";                                                            PROBE[k3]="--workload ground_bunny"
CFG[k4]="--workload k4 --width 3840 --height 2160 --lights 64 --frames 12";     PROBE[k4]="--workload k4 --width 3840 --height 2160 --lights 64"
CFG[k5]="--workload soup --width 2048 --height 2048 --frames 4";                PROBE[k5]="--workload soup --width 2048 --height 2048 --frames 3"
CFG[k5_1080p]="--workload soup --frames 4";                                     PROBE[k5_1080p]="--workload soup --frames 3"
for k in k3 k4 k5 k5_1080p; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${k}_stats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-parity --streams 1 ${CFG[$k]} > $OUT/${k}_s1_bench.json 2> $OUT/${k}_stats.log || echo "stats $k failed"
  cp $(ls $OUT/${k}_stats/*/*kernel_stats.csv | head -1) $OUT/${k}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/${k}_stats
  if [ $k != k3 ]; then
    timeout -k 10 300 python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc ${CFG[$k]} > $OUT/${k}_bench.json 2> $OUT/${k}_bench.log || echo "bench $k failed"
  fi
done
# the headline line as the driver runs it (PMC child passes, parity, cpu_baseline, the soup's secondary measurement)
timeout -k 10 400 python3 $ROOT/bench.py > $OUT/k3_bench.json 2> $OUT/k3_bench.log || echo "bench k3 failed"
G_A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
G_B="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
G_E="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
G_F="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
G_G="TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
for k in k3 k4 k5; do
  bash $ROOT/tools/pmc_groups.sh gpurun_out/${TAG}_profiles/${k}_pmc "${PROBE[$k]}" -- "$G_A" "$G_B" "FETCH_SIZE" "WRITE_SIZE" "$G_E" "$G_F" "$G_G" > $OUT/${k}_pmc.log 2>&1
  mv $OUT/${k}_pmc.txt $OUT/${k}_pmc_summary.txt 2>/dev/null
done
echo collected

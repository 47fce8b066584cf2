#!/bin/bash
# Collect rocprofv3 counters for bench.py, one --pmc pass per invocation (kernel-trace/stats only with
# PMC: gpurun refuses PMC + sys/hip traces).  Usage: profiles/pmc_run.sh <outdir> [bench args...]
# PMC_GROUPS (env, optional): "a" = the issue / wait / HBM groups (default), "s" = adds the scalar-cache groups the packet
# kernels need, "q" = only the three groups bench.py's roofline uses.
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/$OUT"
# Build everything BEFORE any rocprofv3 line: under the profiler the preloaded library initialises the GPU, and a compiler or make
# started from inside the profiled process would be the exec hop this pool forbids (bench.py refuses to build there).
(cd "$ROOT" && python3 -m simple_raytracer_amd.build > /dev/null && python3 -c "from oracle import pyoracle; pyoracle.build()" > /dev/null) || { echo "build failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="FETCH_SIZE"; G4="WRITE_SIZE"; G5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; G6="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
S1="SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAVES"
S2="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB"
# vector-memory side (round 3): what the node-queue kernels wait for
V1="SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
V2="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
V3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
V4="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum"
V5="TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
V6="SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
case "${PMC_GROUPS:-a}" in
  v) GROUPS_=("$V6" "$V1" "$V2" "$V3" "$V4" "$V5") ;;
  q) GROUPS_=("SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE" "$G3" "$G4") ;;
  s) GROUPS_=("$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$S1" "$S2") ;;
  *) GROUPS_=("$G1" "$G2" "$G3" "$G4" "$G5" "$G6") ;;
esac
i=0
for ctrs in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps ${PMC_STEPS:-10} --warmup 2 --no-cpu-baseline --no-pmc --no-parity --no-soup "$@" > "$ROOT/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
echo done

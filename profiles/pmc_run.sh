#!/bin/bash
# Collect rocprofv3 counters for bench.py, one --pmc pass per invocation (kernel-trace/stats only with
# PMC: gpurun refuses PMC + sys/hip traces).  Usage: profiles/pmc_run.sh <outdir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$ROOT/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
echo done

#!/bin/bash
# round-3 session c: new tests (VALU rate, K5 whole frame), valid TA/TCP counter groups, counters of the four node-queue forms
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r3c/tests.log
if [ $rc == 124 ] || [ $rc == 137 ]; then echo "tests timed out"; exit 1; fi
python3 -c "
from simple_raytracer_amd import lib
print('valu_rate', lib.valu_rate(2000), lib.valu_rate(4000))" > gpurun_out/r3c/valu_rate.txt 2>&1
bash tools/pmc_groups.sh gpurun_out/r3c/probe_v42 "--variant 42" -- \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" \
  "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
  "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" 2>&1 | tee gpurun_out/r3c/probe_v42.fail
for v in 0 40 41; do
  bash tools/pmc_groups.sh gpurun_out/r3c/probe_v$v "--variant $v" -- \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES"
done
timeout -k 10 300 python bench.py --variant 42 > gpurun_out/r3c/bench_v42_full.json 2> gpurun_out/r3c/bench_v42_full.err; echo "bench rc=$?"
timeout -k 10 100 python bench.py --variant 43 --no-pmc --no-cpu-baseline --no-soup --no-parity --streams 1 > gpurun_out/r3c/bench_v43.json 2>&1
python3 - <<'PY'
import json
for f in ("bench_v42_full", "bench_v43"):
    try:
        d = json.loads(open(f"gpurun_out/r3c/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["config"]["ms_per_frame"], {k: v["ms"] for k, v in d["kernels"].items()}, d["roofline"].get("frac"), (d["roofline"].get("limiter") or {}).get("name"))
    except Exception as e:
        print(f, "no json", e)
PY

#!/bin/bash
# rocprofv3 --pmc passes of tools/pmc_probe.py, one pass per counter group given as arguments after `--`.
# Usage: tools/pmc_groups.sh <outdir> "<probe args>" -- "<group 1>" "<group 2>" ...   (summary: <outdir>.txt; raw passes are deleted)
OUT=$1; PROBE_ARGS=$2; shift 2; [ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/tools/pmc_probe.py" $PROBE_ARGS > "$ROOT/$OUT/pass$i.log" 2>&1 || echo "pass $i [$ctrs] FAILED: $(grep -i -m2 'error\|invalid\|unknown\|not ' $ROOT/$OUT/pass$i.log | cut -c1-300)"
done
python3 "$ROOT/profiles/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT.txt" 2>&1
rm -rf "$ROOT/$OUT"

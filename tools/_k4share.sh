set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
A="--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity"
python bench.py $A > gpurun_out/st_k4w.json 2>> gpurun_out/st.err
python bench.py --workload ground_bunny --lights 16 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --no-soup > gpurun_out/st_k3L16.json 2>> gpurun_out/st.err
for st in 0 1; do
export SRT_BATCH_STEAL=$st
for sh in 0 1 2 3 4 5 6 7; do
python bench.py $A --emulate-split $sh/8 > gpurun_out/st_k4s${sh}_$st.json 2>> gpurun_out/st.err
done
python bench.py --workload ground_bunny --lights 16 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity --no-soup --emulate-split 3/8 > gpurun_out/st_k3L16s3_$st.json 2>> gpurun_out/st.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/st_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"])
PY

set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "batch or graph or requested_ahead" > gpurun_out/bt_test.log 2>&1 || { tail -30 gpurun_out/bt_test.log; exit 1; }
tail -2 gpurun_out/bt_test.log
A="--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5 --no-pmc --no-cpu-baseline --no-parity"
python bench.py $A --emulate-split 4/8 > gpurun_out/bt_k4s4.json 2>> gpurun_out/bt.err
python bench.py $A --emulate-split 1/8 > gpurun_out/bt_k4s1.json 2>> gpurun_out/bt.err
python bench.py --no-pmc --no-cpu-baseline --no-parity --no-soup --emulate-split 6/8 > gpurun_out/bt_k3s6.json 2>> gpurun_out/bt.err
python bench.py --no-pmc --no-cpu-baseline --no-parity --no-soup --emulate-split 0/2 > gpurun_out/bt_k3h0.json 2>> gpurun_out/bt.err
python bench.py --workload soup --frames 4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-parity --emulate-split 6/8 > gpurun_out/bt_soup6.json 2>> gpurun_out/bt.err
rocprofv3 --kernel-trace --stats -d gpurun_out/bt_prof -o bt -- python bench.py $A --emulate-split 4/8 --streams 1 > gpurun_out/bt_prof.json 2>> gpurun_out/bt.err
python - <<'PY'
import sqlite3, json, glob
for f in sorted(glob.glob("gpurun_out/bt_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"])
for d in sorted(glob.glob("gpurun_out/bt_prof/")):
    db=sqlite3.connect(glob.glob(d+"*.db")[0])
    tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
    for r in db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"):
        if r[0].startswith("_Z") and "at6native" not in r[0]: print("   %-70s n=%4d avg %9.1f us"%(r[0][:70],r[1],r[2]))
PY

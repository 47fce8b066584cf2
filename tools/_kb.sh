cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-pmc --no-parity"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
K4="--workload k4 --width 3840 --height 2160 --lights 64 --frames 8 --steps 5"
for sh in 0 1 2 3; do echo "k4 quarter $sh: streams $(run $K4 --emulate-split $sh/4 --batch off) batch $(run $K4 --emulate-split $sh/4 --batch on)"; done
for sh in 1 4; do echo "k4 eighth $sh: streams $(run $K4 --emulate-split $sh/8 --batch off) batch $(run $K4 --emulate-split $sh/8 --batch on)"; done
echo "k3L16 quarter: streams $(run --no-soup --lights 16 --frames 8 --emulate-split 1/4 --batch off) batch $(run --no-soup --lights 16 --frames 8 --emulate-split 1/4 --batch on)"
echo "k3L16 eighth: streams $(run --no-soup --lights 16 --frames 8 --emulate-split 1/8 --batch off) batch $(run --no-soup --lights 16 --frames 8 --emulate-split 1/8 --batch on)"
echo "k3 half1: streams $(run --no-soup --emulate-split 1/2 --batch off) batch $(run --no-soup --emulate-split 1/2 --batch on)"

# usage: bash scripts/sweep_waves.sh   (on the GPU box) -- A/B of launch geometries through bench.py
export TMPDIR=/tmp
run() { echo "WAVES $1 ROWS $2"; T3MI_WAVES=$1 T3MI_ROWS=$2 timeout -k 10 120 python -u bench.py --cpu-steps 0 --no-profile --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])" || exit 1; }
run 8,4,8,8,8 2,1,2,2,2
run 8,4,8,8,8 4,1,4,2,4
run 8,4,8,8,8 4,1,4,4,4
run 8,4,8,8,8 2,2,2,2,2
run 8,4,8,8,8 2,1,4,2,2
run 8,4,8,8,8 4,1,2,2,2
run 8,4,8,4,8 2,1,2,1,2
run 4,4,4,4,4 2,1,2,2,2
run 4,4,4,4,4 4,1,4,2,4
run 8,4,8,8,8 2,1,2,4,2
run 8,4,8,8,8 2,1,2,2,4

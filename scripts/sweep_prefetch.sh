# Sweep of the L2 prefetcher knobs on the bench workload (run on the GPU box).
# usage: bash scripts/sweep_prefetch.sh "<prefetch lookahead blocks depth>" ...
for cfg in "$@"; do
  set -- $cfg
  printf "PREFETCH=%s LOOKAHEAD=%s BLOCKS=%s DEPTH=%s  " $1 $2 $3 $4
  T3MI_PREFETCH=$1 T3MI_PF_LOOKAHEAD=$2 T3MI_PF_BLOCKS=$3 T3MI_PF_DEPTH=$4 timeout -k 10 120 python bench.py --cpu-steps 0 --no-profile ${BENCH_ARGS} 2>&1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'tok/s', d['hip_event_ms_per_step'], 'ms/step', d['roofline']['frac'])" || exit 1
done

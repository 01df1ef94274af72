# A/B of the pre-attention split norm (down_proj unsplit as producer, next qkv as consumer; T3MI_NORM_SPLIT bit 2), same box, alternating
# usage (GPU box): bash scripts/ln1_ab.sh > gpurun_out/ln1_ab.txt
set -e
run() { # utt steps
  for rep in 1 2; do
    for ns in 1 3; do
      export T3MI_NORM_SPLIT=$ns
      printf "NORM_SPLIT=%d " $ns
      timeout -k 10 120 python scripts/run_session.py --variant llama --utt $1 --s0 64 --steps $2 --reps 2 2>&1 | tail -1
    done
  done
}
run 9 400
run 16 400
run 24 400
run 32 400
run 48 400
run 64 400

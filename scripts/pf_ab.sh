# A/B of the weight prefetch from idle CUs (t3k::PfArgs), same box, alternating: T3MI_PF=0 against the default, whole sessions.
# usage (GPU box): bash scripts/pf_ab.sh > gpurun_out/pf_ab.txt
set -e
run() { # variant utt s0 steps
  for rep in 1 2; do
    for pf in 0 on; do
      if [ $pf = 0 ]; then export T3MI_PF=0; else unset T3MI_PF; fi
      printf "PF=%-3s " $pf
      timeout -k 10 120 python scripts/run_session.py --variant $1 --utt $2 --s0 $3 --steps $4 --reps 2 2>&1 | tail -1
    done
  done
}
run turbo 8 407 300
run turbo 12 407 300
run turbo 16 407 300
run turbo 32 407 300
run llama 3 64 400
run llama 4 64 400
run llama 6 64 400
run llama 8 64 400
run llama 16 64 400
run llama 64 64 400

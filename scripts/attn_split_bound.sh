# Upper bound of a cross-block key split in the many-row attention at few (item, head) blocks:
# the diagnostic library (make diag DIAGFLAGS=-DT3K_DIAG_ATTN_SPLIT) runs S blocks per (item, head), each over 1/S of the
# keys, all writing the same row -- the attention launch's time WITHOUT any merge cost.  Results are wrong; only the time counts.
# usage (GPU box): bash scripts/attn_split_bound.sh > gpurun_out/attn_split_bound.txt
set -e
export T3MI_LIB=$PWD/chatterbox_amd/libt3mi_diag.so
run() { # variant utt s0 steps
  for pf in 0 128,128,63,1,128,16; do
    for S in 0 2 4 8 0; do
      export T3MI_PF=$pf T3MI_DIAG_ATTN_SPLIT=$S
      printf "PF=%-20s split=%d " $pf $S
      timeout -k 10 120 python scripts/run_session.py --variant $1 --utt $2 --s0 $3 --steps $4 --reps 2 2>&1 | tail -1
    done
  done
}
run turbo 8 407 300
run turbo 16 407 300
run llama 4 64 400

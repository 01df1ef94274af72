# A/B of the LayerNorm split (T3MI_NORM_SPLIT) on batched Turbo sessions; usage: bash scripts/ab_turbo_split.sh "<sentence counts>" "<dtypes>"
mkdir -p gpurun_out/ab
export PYTHONPATH=$PWD
for dt in ${2:-bf16}; do
for utt in ${1:-8}; do
for v in 0 1; do
  env T3MI_NORM_SPLIT=$v timeout -k 10 150 python3 scripts/run_session.py --variant turbo --utt $utt --s0 407 --dtype $dt --steps 300 --reps 3 --tag "split=$v" 2>&1 | grep "ms/step" >> gpurun_out/ab/turbo_split.txt || exit 1
done
done
done
cat gpurun_out/ab/turbo_split.txt

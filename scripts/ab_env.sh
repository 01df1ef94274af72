# A/B of one environment switch over batch sizes; usage: bash scripts/ab_env.sh VAR "<values>" "<utts>" [extra run_session args]
mkdir -p gpurun_out/ab
for utt in ${3:-64}; do
for v in $2; do
  env $1=$v timeout -k 10 120 python3 scripts/run_session.py --utt $utt --steps 400 --reps 2 --tag "$1=$v" $4 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab/ab.txt || exit 1
done
done

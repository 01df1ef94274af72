# default dispatch over batch sizes (no switches); usage: bash scripts/ab_default.sh "<utts>" [tag]
mkdir -p gpurun_out/ab
for utt in ${1:-64}; do
  timeout -k 10 120 python3 scripts/run_session.py --utt $utt --steps 400 --reps 2 --tag "${2:-default}" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab/default.txt || exit 1
done

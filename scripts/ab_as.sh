# A/B of the activation-stationary GEMM (T3MI_AS=<mask>) over batch sizes; usage: bash scripts/ab_as.sh "<masks>" "<utts>"
mkdir -p gpurun_out/as
for utt in ${2:-64}; do
for mask in ${1:-0 13}; do
  T3MI_AS=$mask timeout -k 10 120 python3 scripts/run_session.py --utt $utt --steps 400 --reps 2 --tag as$mask 2>&1 | grep -v amdgpu.ids >> gpurun_out/as/ab.txt || exit 1
done
done

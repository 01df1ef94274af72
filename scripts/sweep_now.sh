# ms per step of the standard sessions with the current build: bash scripts/sweep_now.sh [tag]
mkdir -p gpurun_out/sweep
export PYTHONPATH=$PWD
O=gpurun_out/sweep/${1:-now}.txt
: > $O
for u in 4 16 64 128; do timeout -k 10 200 python3 scripts/run_session.py --utt $u --steps 400 --reps 3 --tag llama 2>&1 | grep "ms/step" >> $O || exit 1; done
timeout -k 10 200 python3 scripts/run_session.py --utt 512 --steps 400 --reps 1 --tag llama 2>&1 | grep "ms/step" >> $O || exit 1
for u in 8 16 64; do timeout -k 10 200 python3 scripts/run_session.py --variant turbo --utt $u --s0 407 --steps 300 --reps 3 --tag turbo 2>&1 | grep "ms/step" >> $O || exit 1; done
timeout -k 10 200 python3 scripts/run_session.py --utt 64 --dtype f32 --steps 200 --reps 2 --tag llama-f32 2>&1 | grep "ms/step" >> $O || exit 1
cat $O

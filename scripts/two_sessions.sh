# two concurrent 32-utterance sessions on one GPU against one 64-utterance session (do the launch gaps of one chain
# fill with the other's work?)
mkdir -p gpurun_out/two
timeout -k 10 120 python3 scripts/run_session.py --utt 64 --steps 400 --reps 3 --tag one64 2>&1 | grep -v amdgpu.ids > gpurun_out/two/one64.txt || exit 1
timeout -k 10 120 python3 scripts/run_session.py --utt 32 --steps 400 --reps 6 --tag twoA 2>&1 | grep -v amdgpu.ids > gpurun_out/two/a.txt &
PA=$!
timeout -k 10 120 python3 scripts/run_session.py --utt 32 --steps 400 --reps 6 --tag twoB 2>&1 | grep -v amdgpu.ids > gpurun_out/two/b.txt &
PB=$!
wait $PA || exit 1
wait $PB || exit 1
cat gpurun_out/two/one64.txt gpurun_out/two/a.txt gpurun_out/two/b.txt

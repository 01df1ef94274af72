# Run on the GPU box: kernel-trace stats and PMC traffic passes (one counter per pass, as MI355X_MICROARCH.md prescribes)
# of the bench command (configs[1]) and of the 64-utterance bf16 session (configs[2]).
# usage: bash scripts/profile_round.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
B="python3 bench.py --cpu-steps 0 --no-profile --no-attach"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- $B --steps 100 > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $B --steps 12 --warmup 4 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $B --steps 12 --warmup 4 > $OUT/pmc_write.log 2>&1
python scripts/pmc_traffic.py $OUT/pmc_fetch/run_counter_collection.csv $OUT/pmc_write/run_counter_collection.csv | tee $OUT/traffic.json
S="python3 scripts/run_session.py --utt 64 --s0 64 --dtype bf16"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o run -- $S --steps 400 --chunk 40 > $OUT/c3_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3_fetch -o run -- $S --steps 12 --chunk 12 > $OUT/c3_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c3_write -o run -- $S --steps 12 --chunk 12 > $OUT/c3_write.log 2>&1
python scripts/pmc_traffic.py $OUT/c3_fetch/run_counter_collection.csv $OUT/c3_write/run_counter_collection.csv | tee $OUT/c3_traffic.json
# the same session at the cache length the 400-step benchmark averages (S0 258 + 12 steps: L 258..270)
SL="python3 scripts/run_session.py --utt 64 --s0 258 --dtype bf16"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3L_fetch -o run -- $SL --steps 12 --chunk 12 > $OUT/c3L_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c3L_write -o run -- $SL --steps 12 --chunk 12 > $OUT/c3L_write.log 2>&1
python scripts/pmc_traffic.py $OUT/c3L_fetch/run_counter_collection.csv $OUT/c3L_write/run_counter_collection.csv | tee $OUT/c3L_traffic.json
# Turbo (configs[4]): records and kernel traces of the batched sessions
cd $GRAFT_REPO_ROOT 2>/dev/null || true
bash scripts/turbo_trace.sh round

# Run on the GPU box: kernel-trace stats and PMC traffic passes of the bench command.
# usage: bash scripts/profile_round.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --cpu-steps 0 --no-profile --steps 100 > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 bench.py --cpu-steps 0 --no-profile --steps 12 --warmup 4 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 bench.py --cpu-steps 0 --no-profile --steps 12 --warmup 4 > $OUT/pmc_write.log 2>&1
ls $OUT/*/
python scripts/pmc_traffic.py $OUT/pmc_fetch/run_counter_collection.csv $OUT/pmc_write/run_counter_collection.csv | tee $OUT/traffic.json

#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, one pass each) of the batched Turbo step, 8 sentences at L 407..419, with the weight
# prefetch on (default) and off: gpurun -- 'bash scripts/turbo_pmc.sh'
set -e
R=$PWD; OUT=$R/gpurun_out/turbo_pmc; mkdir -p $OUT
export PYTHONPATH=$R TMPDIR=/tmp
S="python3 $R/scripts/run_session.py --variant turbo --utt 8 --s0 407 --steps 12 --chunk 12"
for pf in on off; do
  if [ $pf = off ]; then export T3MI_PF=0; else unset T3MI_PF; fi
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${pf}_fetch -o run -- $S > $OUT/${pf}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${pf}_write -o run -- $S > $OUT/${pf}_write.log 2>&1
  echo "prefetch $pf"
  python3 $R/scripts/pmc_traffic.py $OUT/${pf}_fetch/run_counter_collection.csv $OUT/${pf}_write/run_counter_collection.csv | tee $OUT/traffic_$pf.json
done

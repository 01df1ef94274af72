#!/bin/bash
# Kernel trace of batched Turbo sessions (8 and 64 sentences, S0 407, 300 steps): gpurun -- 'bash scripts/turbo_trace.sh [tag]'
set -e
R=$PWD; TAG=${1:-run}; OUT=$R/gpurun_out/turbo_$TAG
mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp; export TMPDIR=/tmp
for U in 8 64; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$U -o t -- python3 $R/scripts/run_session.py --variant turbo --utt $U --s0 407 --steps 300 --chunk 40 > $OUT/run$U.log 2>&1
  grep "ms/step" $OUT/run$U.log
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python3 - <<PY
import csv
for u in (8, 64):
    print("sentences", u)
    for r in list(csv.DictReader(open("$OUT/prof%d/t_kernel_stats.csv" % u)))[:8]:
        n = r["Name"].replace("t3k::(anonymous namespace)::", "")[:72]
        print(f'  {n:72s} calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]:>6}%')
PY

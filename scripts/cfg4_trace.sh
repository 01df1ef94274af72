#!/bin/bash
# Kernel trace of the 512-utterance bf16 session (configs[3] at one GPU), 400 steps: gpurun -- 'bash scripts/cfg4_trace.sh'
set -e
R=$PWD; OUT=$R/gpurun_out/cfg4_trace; mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python3 $R/scripts/run_session.py --variant llama --utt 512 --s0 64 --steps 400 --chunk 40 > $OUT/run.log 2>&1
grep "ms/step" $OUT/run.log
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/prof/t_kernel_stats.csv")))[:12]:
    n = r["Name"].replace("t3k::(anonymous namespace)::", "")[:72]
    print(f'  {n:72s} calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]:>6}%')
PY

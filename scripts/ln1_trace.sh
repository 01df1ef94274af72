#!/bin/bash
# Kernel trace of the 64-utterance bf16 session with the pre-attention split norm off (T3MI_NORM_SPLIT=1) and on (3)
set -e
R=$PWD; OUT=$R/gpurun_out/ln1_trace; mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp; export TMPDIR=/tmp
for NS in 1 3; do
  export T3MI_NORM_SPLIT=$NS
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ns$NS -o t -- python3 $R/scripts/run_session.py --variant llama --utt 64 --s0 64 --steps 400 --chunk 40 > $OUT/run$NS.log 2>&1
  grep "ms/step" $OUT/run$NS.log
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python3 - <<PY
import csv
for ns in (1, 3):
    print("T3MI_NORM_SPLIT", ns)
    for r in list(csv.DictReader(open("$OUT/ns%d/t_kernel_stats.csv" % ns)))[:8]:
        n = r["Name"].replace("t3k::(anonymous namespace)::", "")[:72]
        print(f'  {n:72s} calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]:>6}%')
PY

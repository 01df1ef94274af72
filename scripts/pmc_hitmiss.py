"""Per-kernel L2 hit / miss / HBM-read summary from a rocprofv3 --pmc counter_collection.csv.
usage: python scripts/pmc_hitmiss.py <counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in rows:
    name = r["Kernel_Name"]
    if "gemv_kernel<" in name:
        key = "gemv<" + name.split("gemv_kernel<")[1].split(">(")[0].replace("t3k::(anonymous namespace)::", "") + ">"
    else:
        key = re.sub(r"\(t3k::.*|\(float.*|\(void.*|\(unsigned.*|\(int.*", "", name.replace("t3k::(anonymous namespace)::", ""))[-44:]
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
for k, v in sorted(acc.items()):
    n = len(cnt[k])
    print(f"{k:44s} n={n:5d} " + "  ".join(f"{c}={x / n:12.1f}" for c, x in sorted(v.items())))

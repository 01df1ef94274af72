"""Print the top kernels of a rocprofv3 --kernel-trace --stats run.  usage: python scripts/kstats.py <run_kernel_stats.csv> [n]"""
import csv
import re
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(sys.argv[1])))[:n]:
    nm = r["Name"].replace("t3k::(anonymous namespace)::", "")
    nm = re.sub(r"\(t3k::.*|\(float.*|\(void.*|\(unsigned.*", "", nm)
    print(f"  {nm:66s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:8.2f} us  {float(r['Percentage']):5.1f}%")

"""Mean per-dispatch counter values per kernel from rocprofv3 --pmc CSVs (tools/pmc_dw.sh).  usage: pmc_dw_sum.py <dir>"""
import csv, glob, sys, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(k)
    for c in sorted(d[k]):
        v = d[k][c]
        print(f"    {c:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")

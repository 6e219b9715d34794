"""Side-by-side per-kernel averages of two rocprofv3 kernel-stats CSVs (same box, same bench command).  usage: kstats_diff.py old.csv new.csv [calls_per_forward_divisor]"""
import csv, sys, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"\(.*$", "", n).replace("void ", "")
        d[n] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
nf = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
keys = sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0, 0))[1] + a.get(k, (0, 0, 0))[1]))
ta = tb = 0
print(f"{'kernel':60s} {'calls/fwd':>9s} {'old us':>9s} {'new us':>9s} {'d/fwd us':>9s}")
for k in keys:
    ca, tota, ava = a.get(k, (0, 0, 0)); cb, totb, avb = b.get(k, (0, 0, 0))
    if max(tota, totb) / nf < 3: continue
    ta += tota / nf; tb += totb / nf
    print(f"{k[:60]:60s} {max(ca,cb)/nf:9.1f} {ava:9.1f} {avb:9.1f} {(totb-tota)/nf:9.1f}")
print(f"{'sum per forward (us)':60s} {'':9s} {ta:9.1f} {tb:9.1f} {tb-ta:9.1f}")

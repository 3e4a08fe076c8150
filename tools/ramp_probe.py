"""Kernel durations and idle gaps of the first steps after a synchronisation against later ones, from a rocprofv3 kernel trace of
tools/resample_loop.py-like runs: python tools/ramp_probe.py TRACE_DIR"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0][-40:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps = runs [k_scan, k_search_multi, k_step]; group by k_step occurrences after the first 10 kernels
steps, cur = [], []
for name, s, e in ks:
    cur.append((name, s, e))
    if "k_step" in name:
        steps.append(cur); cur = []
def summary(group):
    d = {}
    for st in group:
        for name, s, e in st:
            key = "k_step" if "k_step" in name else "k_scan" if "k_scan" in name else "k_search" if "k_search" in name else name[-20:]
            d.setdefault(key, []).append((e - s) / 1e3)
    span = [(st[-1][2] - st[0][1]) / 1e3 for st in group]
    period = [(b[0][1] - a[0][1]) / 1e3 for a, b in zip(group, group[1:])]
    return {k: round(sum(v) / len(v), 2) for k, v in d.items()}, round(sum(span) / len(span), 2), round(sum(period) / max(1, len(period)), 2)
n = len(steps)
for lo, hi in ((2, 12), (12, 32), (32, 100), (n - 200, n - 1)):
    if hi <= n and lo >= 0:
        print(f"steps {lo}-{hi}:", *summary(steps[lo:hi]))

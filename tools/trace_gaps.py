"""print a window of a rocprofv3 kernel trace: kernel, duration, idle gap before it (us)"""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mid = int(len(rows) * float(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2)
prev = None
for r in rows[mid:mid + int(sys.argv[3]) if len(sys.argv) > 3 else mid + 30]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{r['Kernel_Name'][:70]:70s} dur={(e - s) / 1e3:7.1f} gap={gap:7.1f}")
    prev = e

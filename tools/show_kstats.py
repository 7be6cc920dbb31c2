"""Per-step table of a rocprofv3 kernel_stats.csv. Usage: show_kstats.py <csv> [steps=23] [rows=25]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 23
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:72]:72s} calls {r['Calls']:>6s} us/step {float(r['TotalDurationNs'])/1e3/steps:8.1f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
print(f"total kernel time per step: {tot/1e3/steps:.1f} us")

"""Histogram of GPU idle gaps over the whole steady-state part of a kernel trace, and where (next kernel) they occur.
Usage: trace_gap_hist.py <run_kernel_trace.csv> [skip first fraction=0.5]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * skip):]
short = lambda s: re.sub(r"\(.*", "", s.replace("void ", "").replace("sm::", ""))[:40]
busy_end, busy = ev[0][0], 0
hist, where = collections.Counter(), collections.Counter()
tot_gap = 0
for s, e, name in ev:
    if s > busy_end:
        g = s - busy_end
        tot_gap += g
        b = "<20us" if g < 20e3 else "<100us" if g < 100e3 else "<1ms" if g < 1e6 else "<5ms" if g < 5e6 else ">=5ms"
        hist[b] += g
        if g >= 1e6:
            where[short(name)] += 1
    busy += max(0, e - max(s, busy_end))
    busy_end = max(busy_end, e)
wall = busy_end - ev[0][0]
steps = sum(1 for e in ev if "step_begin_kernel" in e[2])
print(f"{steps} steps, wall {wall/1e6:.1f} ms = {wall/1e3/max(steps,1):.1f} us/step, busy {busy/1e3/max(steps,1):.1f} us/step, idle {tot_gap/1e3/max(steps,1):.1f} us/step")
print("idle time by gap length (us/step):", {k: round(v / 1e3 / max(steps, 1), 1) for k, v in hist.items()})
print("gaps >= 1 ms occur before:", where.most_common(12))

"""Idle gaps and per-stream busy time of the steady-state steps of a rocprofv3 kernel trace.
Usage: trace_gaps.py <run_kernel_trace.csv> [first step from the end=30] [steps=6]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
qk = "Queue_Id" if "Queue_Id" in rows[0] else None
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qk, "?") if qk else "?") for r in rows)
head = [i for i, e in enumerate(ev) if "step_begin_kernel" in e[2]]
seg = ev[head[-back]:head[-back + n]]
t0 = seg[0][0]
short = lambda s: re.sub(r"\(.*", "", s.replace("void ", "").replace("sm::", ""))[:44]
busy_end, busy, gaps = t0, 0, []
per_q = collections.Counter()
for s, e, name, q in seg:
    if s > busy_end and s - busy_end > 15000:
        gaps.append((busy_end - t0, s - busy_end, short(name)))
    busy += max(0, e - max(s, busy_end))
    busy_end = max(busy_end, e)
    per_q[q] += e - s
wall = seg[-1][1] - t0
print(f"{n} steps: wall {wall/1e3/n:.1f} us/step, GPU busy (union) {busy/1e3/n:.1f} us/step, kernel time per queue (us/step): "
      + ", ".join(f"q{q}: {v/1e3/n:.1f}" for q, v in per_q.items()))
print("idle gaps > 15 us (at, length, next kernel):")
for at, ln, nm in gaps:
    print(f"   {at/1e3:9.1f}  {ln/1e3:7.1f}  {nm}")
print("--- first step, kernel by kernel (start, dur, queue, name)")
for s, e, name, q in ev[head[-back]:head[-back + 1]]:
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f}  q{q}  {short(name)}")

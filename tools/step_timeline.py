"""Timeline of ONE steady-state step from a rocprofv3 kernel trace (run_kernel_trace.csv): phases, gaps, overlap.
Usage: step_timeline.py <run_kernel_trace.csv> [step_index_from_end=3]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# a step starts with the step-head kernel (older traces: with the launches following the update kernel)
head = [i for i, e in enumerate(ev) if "step_begin_kernel" in e[2]]
if len(head) > k + 1:
    step = ev[head[-k - 1]:head[-k]]
else:
    adam = [i for i, e in enumerate(ev) if "adam_kernel<true>" in e[2]]
    a0, a1 = adam[-k - 1], adam[-k]
    step = ev[a0 + 1:a1 + 1]
t0 = step[0][0]
def short(n):
    n = n.replace("void ", "").replace("sm::", "")
    return re.sub(r"\(.*", "", n)[:46]
busy_end, busy = t0, 0
print(f"step of {len(step)} kernels, wall {(step[-1][1]-t0)/1e3:.1f} us")
last_end = t0
for s, e, n in step:
    if s > busy_end:
        gap = s - busy_end
        if gap > 3000:
            print(f"      --- GPU idle {gap/1e3:6.1f} us")
    busy += max(0, e - max(s, busy_end))
    busy_end = max(busy_end, e)
for s, e, n in step:
    print(f"{(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}  {short(n)}")
print(f"busy (union) {busy/1e3:.1f} us of wall {(step[-1][1]-t0)/1e3:.1f} us")

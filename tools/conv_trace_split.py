"""Split the conv rows of a rocprofv3 kernel trace into the STEP's launches and the launches of set_style_image() /
set_view() (style pyramid, content target), which the --stats summary averages together.

A launch belongs to a step when it runs on the trunk's queue (the queue of the ``step_begin_kernel`` launches) between a
``step_begin_kernel`` and the ``adam_kernel`` that closes that step. (Rounds 1-4 classified by DURATION - grouped launches
over four UV levels last > 100 us - which cannot tell a one-level step's 16-39 us launches from the set-up passes': every
c2 step launch was labelled 'set_style_image / set_view', VERDICT r4 weak #9.)
Usage: conv_trace_split.py <run_kernel_trace.csv> <out.csv>"""
import collections
import csv
import sys

src, out = sys.argv[1:3]
rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
qkey = "Queue_Id" if rows and "Queue_Id" in rows[0] else None
trunk = None
if qkey:
    c = collections.Counter(r[qkey] for r in rows if "step_begin_kernel" in r["Kernel_Name"])
    trunk = c.most_common(1)[0][0] if c else None
acc = collections.defaultdict(lambda: [0, 0.0])
in_step = False
for r in rows:
    name = r["Kernel_Name"]
    on_trunk = trunk is None or r[qkey] == trunk
    if "step_begin_kernel" in name and on_trunk:
        in_step = True
    is_conv = "conv3x3_split_kernel" in name or "conv3x3_mfma_kernel" in name or "conv_tail_" in name
    if is_conv:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        kind = "step" if (in_step and on_trunk) else "set_style_image / set_view"
        family = "ALL tail second passes" if "conv_tail_" in name else ("ALL split conv" if "split" in name else "ALL fp32 conv")
        for key in ((name.split("(")[0], kind), (family, kind)):
            acc[key][0] += 1
            acc[key][1] += us
    if "adam_kernel" in name and on_trunk:
        in_step = False
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launch class", "launches", "total_us", "avg_us"])
    for (name, kind), (n, t) in sorted(acc.items()):
        w.writerow([name, kind, n, round(t, 1), round(t / n, 2)])
print(open(out).read())

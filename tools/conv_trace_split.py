"""Split the conv rows of a rocprofv3 kernel trace into the step's grouped launches (all UV levels in one launch) and
the small single-image launches of set_style_image() / set_view() (style pyramid, content target), which the
--stats summary averages together. Usage: conv_trace_split.py <run_kernel_trace.csv> <out.csv> [threshold_us=100]"""
import csv, sys, collections
src, out = sys.argv[1:3]
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(src)):
    name = r["Kernel_Name"]
    if "conv3x3_split_kernel" not in name and "conv3x3_mfma_kernel" not in name:
        continue
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    kind = "step (grouped over the UV levels)" if us >= thr else "set_style_image / set_view (single image)"
    for key in ((name.split("(")[0], kind), ("ALL split conv" if "split" in name else "ALL fp32 conv", kind)):
        acc[key][0] += 1
        acc[key][1] += us
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launch class", "launches", "total_us", "avg_us"])
    for (name, kind), (n, t) in sorted(acc.items()):
        w.writerow([name, kind, n, round(t, 1), round(t / n, 2)])
print(open(out).read())

"""Condense rocprofv3 --pmc counter_collection CSVs (gpurun_out/pmc/*/) into one small table:
per kernel variant + grid size: launches, total ns, summed counters. Usage: summarize_pmc.py <pmc_dir> <out.csv>"""
import collections, csv, glob, os, sys

src, out = sys.argv[1], sys.argv[2]
table = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob(os.path.join(src, "*/"))):
    cc = os.path.join(d, "run_counter_collection.csv")
    if not os.path.exists(cc):
        continue
    trace = {t["Dispatch_Id"]: t for t in csv.DictReader(open(os.path.join(d, "run_kernel_trace.csv")))}
    seen = set()
    for x in csv.DictReader(open(cc)):
        name = x["Kernel_Name"]
        if not name.startswith("void sm::") and not name.startswith("sm::"):
            continue
        short = name.split("(")[0].replace("void ", "")
        key = (short, x["Grid_Size"])
        table[key][x["Counter_Name"]] += float(x["Counter_Value"])
        tag = (os.path.basename(d.rstrip("/")), x["Dispatch_Id"])
        if tag not in seen:
            seen.add(tag)
            t = trace[x["Dispatch_Id"]]
            table[key]["ns@" + tag[0][:24]] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
            table[key]["launches@" + tag[0][:24]] += 1
cols = sorted({c for v in table.values() for c in v})
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads"] + cols)
    for k in sorted(table):
        w.writerow(list(k) + ["%.6g" % table[k].get(c, 0) for c in cols])
print("wrote", out, len(table), "rows")

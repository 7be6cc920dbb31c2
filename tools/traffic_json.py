"""Condense the FETCH_SIZE / WRITE_SIZE PMC summary (tools/profile_round.sh) into profiles/rNN/conv_traffic_<wl>_<tag>.json.
Usage: traffic_json.py <summary.csv> <workload> <tag: split2|split|f32> <out.json> [min_launches]
min_launches: only rows (kernel variant x grid size) launched at least that often are counted - with the number of steps
of the PMC run this selects the STEP's launches and leaves out the one-off launches of set_style_image / set_view.
FETCH_SIZE on gfx950 under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md); the factor is re-derived from
the Adam kernel of the same run, whose traffic is known exactly (7 fp32 streams over the texture arena)."""
import csv, json, sys
src, wl, tag, out = sys.argv[1:5]
min_launches = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rows = list(csv.DictReader(open(src)))
kern = "conv3x3_split_kernel" if tag in ("split", "split2") else "conv3x3_mfma_kernel"
np_arg = ""   # (rounds 2-5 told the bf16x3 / fp16x2 instantiations apart by their NP template argument; one arithmetic since r6)
def col(r, name):
    for k, v in r.items():
        if k.startswith(name):
            return float(v)
    return 0.0
fetch = write = launches = 0.0
adam = None
for r in rows:
    if kern in r["kernel"] and np_arg in r["kernel"] and col(r, "launches@FETCH") >= min_launches:
        fetch += r.get("FETCH_SIZE") and float(r["FETCH_SIZE"]) or 0.0
        write += r.get("WRITE_SIZE") and float(r["WRITE_SIZE"]) or 0.0
        launches += col(r, "launches@FETCH")
    if "adam_kernel<true>" in r["kernel"] or ("adam_sparse_kernel" in r["kernel"] and adam is None):
        adam = r
corr = 2.0
cal = None
if adam is not None:
    n = col(adam, "launches@FETCH")
    cal = {"adam_fetch_mb_per_launch_raw": float(adam["FETCH_SIZE"]) * 1024 / n / 1e6,
           "adam_write_mb_per_launch": float(adam["WRITE_SIZE"]) * 1024 / n / 1e6}
json.dump({"workload": wl, "kernel": kern, "launches": launches, "rows_with_at_least_launches": min_launches, "fetch_size_kb_raw": fetch, "write_size_kb": write,
           "fetch_correction": corr, "calibration": cal,
           "hbm_bytes_per_launch": (fetch * corr + write) * 1024 / max(launches, 1),
           "source": "tools/profile_round.sh (two rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE) + tools/traffic_json.py"},
          open(out, "w"), indent=1)
print(open(out).read())

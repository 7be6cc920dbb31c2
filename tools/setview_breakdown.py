"""GPU work of a view change, kernel by kernel. Two modes:
  run   <workload> [n_views]     (under rocprofv3 --kernel-trace): set_view alone on several views, windows 5 ms apart
  parse <run_kernel_trace.csv>   : per window wall / busy time, and the kernels of the LAST windows aggregated
"""
import sys, os, time, csv, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(workload, n_views):
    import torch
    import bench as B
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import StepEngine
    wl = B.WORKLOADS[workload]
    eng = StepEngine(B.engine_config(wl), S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, *B.STYLE_HW))
    eng.prepare_ahead = False
    seeds = [0, 2, 6, 7, 9, 11, 12, 14][:n_views]
    views = [B.to_device(v, "cuda") for v in B.make_views(wl, seeds)]
    for v in views[:2]:
        eng.training_step(v)
    torch.cuda.synchronize()
    host = []
    for v in views * 2:
        time.sleep(0.005)
        t0 = time.perf_counter()
        eng.set_view(v)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        host.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
    print("set_view host ms / until-done ms:", [(round(a, 2), round(b, 2)) for a, b in host])
    print("marks (ms per call):", {k: round(1e3 * v / max(eng.__dict__.get("set_view_calls", len(host) + 2), 1), 3) for k, v in eng.__dict__.get("set_view_marks", {}).items()})


def parse(path, last=4):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    windows, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur) > 2_000_000:
            windows.append(cur)
            cur = []
        cur.append(e)
    windows.append(cur)
    short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("sm::", "").replace("at::native::", ""))[:70]
    agg = collections.OrderedDict()
    for k, w in enumerate(windows):
        wall = (max(x[1] for x in w) - w[0][0]) / 1e3
        busy = sum(x[1] - x[0] for x in w) / 1e3
        print(f"window {k}: {len(w)} kernels, wall {wall:.1f} us, kernel time {busy:.1f} us")
    for w in windows[-last:]:
        for s, e, n in w:
            a = agg.setdefault(short(n), [0, 0.0])
            a[0] += 1
            a[1] += (e - s) / 1e3
    print(f"--- kernels of the last {last} windows (per window) ---")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / last:9.1f} us  x{c / last:5.1f}  {n}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    else:
        parse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)

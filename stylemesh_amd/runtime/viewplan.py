"""The per-view constants of a view through TWO grouped library calls (``sm_view_masks`` / ``sm_view_lists``) instead of
one call per (level, layer): the host side of include/stylemesh_hip.h's "per-view constants of a whole view" section.

What it computes is what ``StepEngine._set_view_body`` / ``_finish_view`` and ``sparsity.build_tile_lists`` compute - level
masks and pixel weights (reference model/model.py:204-254), layer-resolution masks, counts and level factors
(content_and_style_losses.py:146-217), the content targets' resizes, the dead-tile analysis and every active list of the
step's launches - into the same persistent per-slot buffers, so the step's kernels, problem tables and captured graphs
never notice. What changes is the cost of a view change on the HOST: 130 launches for a one-level view, 250 for four
levels (2.6 / 4.3 ms of interpreter + launch time, measured in round 4) against two ``ctypes`` calls whose descriptors are
built once per (slot, view shape) and only get their input pointers refreshed. That is what a schedule that changes the
view every step is bound by (``--index_repeat 1``: scripts/train/optimize_texture_scannet_dip.sh:16,
data/abstract_dataset.py:498-512).

The one read-back of a view change (list lengths = grid sizes, mask sums = the empty-level filter of model/model.py:256-257)
is a single small device-to-host copy into pinned memory; ``PendingView.finish`` waits for it as late as the caller
allows (a view prepared one ahead: at the swap).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import hip, ops
from .fmap import FMap
from .vgg import NODES, POOL_OUTPUT, PRE_POOL, depth_of, fuse_pool_fwd, layer_hw


def resident_lists() -> bool:
    """Do the 64-output-channel launches of the fp16x2 mode take QUAD lists (the resident-input kernel, SM_LIST_QUADS)?"""
    import os
    return ops.CONV_MODE == "split2" and os.environ.get("STYLEMESH_RESIDENT", "1") != "0"


class TileLists(dict):
    """{job key: (list tensor, live fraction)} of a view; ``quads``: the keys whose lists hold vertical quads of segments."""
    quads = frozenset()


def list_jobs(deepest: str, pairs: bool, resident: bool = False):
    """The step's active-list consumers in launch order: (key, layer name the list is built from, mode, bn, group,
    pair layer) - the job table of ``sparsity.build_tile_lists``. mode 0: free segments, 1: segment pairs over the POOLED
    layer's need map, 2: aligned tiles of bn positions; ``resident``: the launches with 64 output channels and whole
    64-channel input phases (conv1_2 forward / data gradient, conv2_1's data gradient) take QUADS instead - mode 3 over the
    pooled layer's need map, mode 4 over the layer's own (``sm_cover_problem::quad``), four entries per tile."""
    jobs = []
    names = {"img"} | {n[2] for n in NODES[:depth_of(deepest) + 1]}
    for kind, src, dst, cin, cout in NODES[:depth_of(deepest) + 1]:
        if kind == "pool":
            jobs.append((("pool", dst), dst, 2, ops.plane_tile_positions(1), 1, None))
            continue
        bn, group = ops.conv_list_format(4 if cin == 3 else cin, cout)
        quads_f = resident and group > 0 and cout == 64 and cin % 64 == 0
        if pairs and group > 0 and dst in PRE_POOL and POOL_OUTPUT[dst] in names:
            jobs.append(((kind, "fp"), POOL_OUTPUT[dst], 3, 32, 4, dst) if quads_f else ((kind, "fp"), POOL_OUTPUT[dst], 1, 32, group, dst))
        elif group > 0:
            jobs.append(((kind, "f"), dst, 4, 32, 4, None) if quads_f else ((kind, "f"), dst, 0, 32, group, None))
        else:
            jobs.append(((kind, "f"), dst, 2, bn, 1, None))
        if src != "img":
            bn, group = ops.conv_list_format(cout, cin)
            if resident and group > 0 and cin == 64 and cout % 64 == 0:
                jobs.append(((kind, "b"), src, 4, 32, 4, None))
                continue
            jobs.append(((kind, "b"), src, 0, 32, group, None) if group > 0 else ((kind, "b"), src, 2, bn, 1, None))
        else:
            jobs.append((("img", "d"), "img", 2, ops.plane_tile_positions(0), 1, None))
    return jobs


class PendingView:
    """A view whose launches are enqueued and whose read-back is on its way."""

    def __init__(self, plan, event, reducer_count=None):
        self.plan, self.event, self.reducer_count = plan, event, reducer_count

    def finish(self):
        """Wait for the read-back; -> (view_tiles dict, mask sums per level of the plan's level table)."""
        self.event.synchronize()
        return self.plan.parse_summary()


class ResidentView:
    """The per-view state of a view that stays in HBM after its first visit (round 5): everything
    ``set_view`` computes depends on the view only - level maps, layer masks and factors, content targets, active lists,
    the sorted scatter plan, touch flags - and the reference's schedules visit the same views in every epoch
    (data/abstract_dataset.py:498-512). A revisit copies the state back into the slot's fixed-address buffers (a few
    device-to-device copies) instead of recomputing it (69 / 96 kernels, 0.65 / 2.2 ms of GPU work for one / four levels)."""

    def __init__(self, plan, scatter_plan, view_flags, active, max_bytes=None):
        """``max_bytes``: what is left of the caller's budget - a view that would not fit raises ``MemoryError`` BEFORE it
        allocates (the budget used to be checked against what was already kept, and could be overshot by a view)."""
        self.plan_key = plan.cache_key
        self.active = tuple(active)
        # lists: only the live prefix of every list buffer (its length is in the summary)
        ints = plan.summary_host[:9 * len(plan.specs)].view(len(plan.specs), 9)[:, 0].tolist() if plan.specs else []
        list_len = {id(sp["out"]): n for sp, n in zip(plan.specs, ints)}
        src = [t[:list_len[id(t)]] if id(t) in list_len else t for t in plan.outputs]
        self.n_plan = len(src)
        self.scatter_meta = None
        if scatter_plan is not None and scatter_plan.level_hw is not None:
            sp, n = scatter_plan, scatter_plan.n_entries
            self.scatter_meta = (n, list(sp.level_hw), sp.sorted_in)
            src += [sp.bufs[sp.sorted_in][:n], sp.bufs[2 + sp.sorted_in][:n],
                    sp.bufs[5][:hip.lib.sm_tex_scatter_plan_cross_bytes(n)]]
        self.has_flags = view_flags is not None
        if self.has_flags:
            src.append(view_flags)
        # ONE allocation per view (a device allocation synchronises: forty of them per new view would cost milliseconds)
        sizes = [(t.numel() * t.element_size() + 255) // 256 * 256 for t in src]
        if max_bytes is not None and sum(sizes) > max_bytes:
            raise MemoryError(f"resident view of {sum(sizes)} bytes against {int(max_bytes)} left of the budget")
        self.flat = torch.empty(max(sum(sizes), 256), dtype=torch.uint8, device=src[0].device)
        self.tensors, off = [], 0
        for t, sz in zip(src, sizes):
            self.tensors.append(self.flat[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape))
            off += sz
        torch._foreach_copy_(self.tensors, src)
        self.ready = torch.cuda.Event()     # (a later visit may restore on another stream)
        self.ready.record()
        self.summary_host = plan.summary_host.clone()
        self.nbytes = self.flat.numel()

    def restore(self, plan, scatter_plan, view_flags):
        """Enqueue the copies back into the slot's buffers (current stream): the plan's outputs, the scatter plan, the
        view's touch flags."""
        torch.cuda.current_stream().wait_event(self.ready)
        src = list(self.tensors)
        dsts = [d[:t.numel()] if d.dim() == 1 and d.numel() != t.numel() else d for d, t in zip(plan.outputs, src[:self.n_plan])]
        k = self.n_plan
        if self.scatter_meta is not None:
            n, level_hw, sorted_in = self.scatter_meta
            scatter_plan._ensure(n, [tuple(x) for x in level_hw])
            scatter_plan.sorted_in = sorted_in
            dsts += [scatter_plan.bufs[sorted_in][:n], scatter_plan.bufs[2 + sorted_in][:n], scatter_plan.bufs[5][:src[k + 2].numel()]]
            k += 3
        if self.has_flags:
            dsts.append(view_flags)
        torch._foreach_copy_(dsts, src[:len(dsts)])


class CachedPending:
    """``PendingView`` of a resident view: nothing to wait for - the list lengths and mask sums are the stored summary."""

    def __init__(self, plan, resident):
        self.plan, self.resident, self.event, self.reducer_count, self.reducer = plan, resident, None, None, None

    def finish(self):
        self.plan.summary_host.copy_(self.resident.summary_host)
        return self.plan.parse_summary()


class ViewPlan:
    """Persistent buffers + cached descriptors of one (slot, view shape, active level set)."""

    def __init__(self, eng, slot: int, h: int, w: int, level_hw, maps_levels, active_levels):
        """``level_hw``: (H, W) of every UV level of the batch; ``maps_levels``: indices of the levels that get maps (all
        with depth scaling, the last one without); ``active_levels``: the levels assumed non-empty (subset of maps_levels)."""
        cfg, dev = eng.cfg, eng.device
        self.eng, self.slot, self.h, self.w = eng, slot, h, w
        self.level_hw, self.maps_levels, self.active = list(level_hw), list(maps_levels), list(active_levels)
        # what a resident view's state is valid for: this plan's shape, whatever the slot
        self.cache_key = (h, w, tuple(level_hw), tuple(active_levels), ops.CONV_MODE, tuple(eng.injected),
                          fuse_pool_fwd(), resident_lists(), bool(cfg.use_depth_scaling),
                          bool(cfg.use_angle_weight), float(cfg.angle_threshold), tuple(cfg.content_layers or ()))
        n_levels = len(level_hw)
        depth = bool(cfg.use_depth_scaling)
        assert eng._wslot == slot, "a plan is built while its slot is the one being written"

        def persist(key, factory):
            return eng._persist(("plan",) + key, factory)
        self._persist = persist
        # ---- masks descriptor -------------------------------------------------------------------------------
        d = self.masks_desc = hip.ViewMasksDesc()
        d.h, d.w, d.angle_threshold, d.n_levels = h, w, float(cfg.angle_threshold), n_levels
        self.E = persist(("E", n_levels if depth else 1, h, w), lambda: torch.empty(n_levels if depth else 1, h, w, device=dev))
        self.Wt = persist(("Wt", n_levels, h, w), lambda: torch.empty(n_levels, h, w, device=dev)) if depth else None
        d.E, d.Wt = self.E.data_ptr(), (self.Wt.data_ptr() if depth else None)
        n_ll = len(eng.loss_layers)
        self.consts = eng._persist(("consts", n_levels, n_ll), lambda: torch.zeros(n_levels, n_ll, 4, device=dev))
        # ---- lists: jobs -> unique list specs ----------------------------------------------------------------
        self.jobs = list_jobs(eng.deepest, fuse_pool_fwd(), resident_lists())
        layer_names = ["img"] + [n[2] for n in NODES[:depth_of(eng.deepest) + 1]]
        self.layer_index = {n: i for i, n in enumerate(layer_names)}
        n_specs = len({(j[1], j[2], j[3], j[4], j[5]) for j in self.jobs})
        # summary: 9 ints per list, then one float per level (the mask sums), then one spare int
        self.summary = persist(("summary", n_specs, n_levels), lambda: torch.zeros(9 * n_specs + n_levels + 1, dtype=torch.int32,
                                                                                 device=dev))
        self.summary_host = persist(("summary_host", n_specs, n_levels),
                                    lambda: torch.zeros(9 * n_specs + n_levels + 1, dtype=torch.int32).pin_memory())
        self.msums = self.summary[9 * n_specs:9 * n_specs + n_levels].view(torch.float32)
        # every device buffer a STEP (or the view's activation) reads of what this plan computes - the state a resident
        # view keeps (``ResidentView``): level maps, layer masks, constants, content targets, active lists
        self.outputs = [self.consts]
        self.levels = []
        want_pw = cfg.use_angle_weight or cfg.use_depth_scaling
        for i, (H, W) in enumerate(level_hw):
            L = d.levels[i]
            L.H, L.W, L.has_maps = H, W, int(i in self.maps_levels)
            rec = {"index": i, "H": H, "W": W}
            if L.has_maps:
                hw = (i, H, W)
                rec["M"] = eng._persist(("M",) + hw, lambda: torch.empty(H, W, device=dev))
                rec["pixel_weight"] = eng._persist(("pw",) + hw, lambda: torch.empty(H, W, device=dev)) if want_pw else None
                rec["passed"] = eng._persist(("passed",) + hw, lambda: torch.empty(H, W, dtype=torch.uint8, device=dev))
                L.M, L.passed = rec["M"].data_ptr(), rec["passed"].data_ptr()
                L.pixel_weight = rec["pixel_weight"].data_ptr() if want_pw else None
                L.m_sum = self.msums[i:i + 1].data_ptr()
                self.outputs += [t for t in (rec["M"], rec["pixel_weight"], rec["passed"]) if t is not None]
            self.levels.append(rec)
        masks = []
        for a in self.active:
            rec = self.levels[a]
            rec["masks"], rec["counts"], rec["factor"] = {}, {}, {}
            for k, layer in enumerate(eng.loss_layers):
                hl, wl = layer_hw(layer, rec["H"], rec["W"])
                m = eng._persist(("lmask", a, layer, hl, wl), lambda: FMap(3, hl, wl, dev))
                self.outputs.append(m.buf)
                rec["masks"][layer], rec["counts"][layer] = m, self.consts[a, k, 0:3]
                rec["factor"][layer] = self.consts[a, k, 3:4]
                masks.append(hip.ViewLayerMask(a, k, hl, wl, m.ptr, self.consts[a, k, 0:3].data_ptr(),
                                               self.consts[a, k, 3:4].data_ptr()))
        self.masks_arr = (hip.ViewLayerMask * max(len(masks), 1))(*masks)
        d.n_masks, d.masks = len(masks), C.cast(self.masks_arr, C.POINTER(hip.ViewLayerMask))
        resizes = []
        self.content_bufs = None
        if cfg.content_layers and self.active:
            from .vgg import LevelBuffers
            key = (h, w)
            if key not in eng._content_bufs:
                eng._content_bufs[key] = LevelBuffers(h, w, eng.deepest_content, False, dev)
            self.content_bufs = cb = eng._content_bufs[key]
            for a in self.active:
                rec = self.levels[a]
                rec["content_target"] = {}
                for layer in cfg.content_layers:
                    src = cb.act[layer]
                    hl, wl = layer_hw(layer, rec["H"], rec["W"])
                    dst = eng._persist(("ctarget", a, layer, hl, wl), lambda: FMap(src.C, hl, wl, dev))
                    rec["content_target"][layer] = dst
                    self.outputs.append(dst.buf)
                    resizes.append(hip.ViewResize(src.ptr, src.C, src.H, src.W, dst.ptr, hl, wl))
        self.resize_arr = (hip.ViewResize * max(len(resizes), 1))(*resizes)
        d.n_resizes, d.resizes = len(resizes), C.cast(self.resize_arr, C.POINTER(hip.ViewResize))
        # ---- lists descriptor --------------------------------------------------------------------------------
        self.lists_desc = None
        self.specs, self.job_spec = [], {}
        if eng.sparse_tiles and self.active and eng.deepest is not None:
            self._build_lists_desc(layer_names)

    # ------------------------------------------------------------------------------------------------------------
    def _build_lists_desc(self, layer_names):
        eng, dev = self.eng, self.eng.device
        L = self.lists_desc = hip.ViewListsDesc()
        act = self.active
        L.n_levels, L.n_layers = len(act), len(layer_names)
        nodes = NODES[:depth_of(eng.deepest) + 1]
        injected = set(eng.injected)
        for j, (kind, src, out, _, _) in enumerate(nodes):
            L.node_is_pool[j] = int(kind == "pool")
            L.node_src[j] = self.layer_index[src]
        for name, idx in self.layer_index.items():
            L.injected[idx] = int(name in injected)
        self.need = []
        for g, a in enumerate(act):
            rec = self.levels[a]
            H, W = rec["H"], rec["W"]
            L.M[g], L.H[g], L.W[g] = rec["M"].data_ptr(), H, W
            dims = {"img": (H, W)}
            for _, _, out, _, _ in nodes:
                dims[out] = layer_hw(out, H, W)
            total = sum(hh * ww for hh, ww in dims.values())
            arena = self._persist(("need", a, H, W, len(layer_names)), lambda: torch.empty(total, device=dev))
            off, per = 0, {}
            for name in layer_names:
                hh, ww = dims[name]
                per[name] = arena[off:off + hh * ww].view(hh, ww)
                idx = self.layer_index[name]
                L.need[g][idx], L.lh[g][idx], L.lw[g][idx] = per[name].data_ptr(), hh, ww
                off += hh * ww
            self.need.append(per)
        # unique list specs
        specs = []
        for key, layer, mode, bn, group, pair_layer in self.jobs:
            sk = (layer, mode, bn, group, pair_layer)
            if sk not in self.job_spec:
                self.job_spec[sk] = len(specs)
                specs.append(sk)
        arr = (hip.ViewList * len(specs))()
        shapes = tuple((self.levels[a]["H"], self.levels[a]["W"]) for a in act)
        for s, (layer, mode, bn, group, pair_layer) in enumerate(specs):
            caps, n_all = [], 0
            for g in range(len(act)):
                hh, ww = self.need[g][layer].shape
                if mode == 1:
                    fh, fw = self.need[g][pair_layer].shape
                    caps.append(2 * hh * ((ww + 15) // 16 + 1) + 2)
                    n_all += (fh * hip.row_stride(fw) + 31) // 32
                elif mode == 3:     # quads over the pooled need map: two pooled rows per group, four entries per run
                    fh, fw = self.need[g][pair_layer].shape
                    caps.append(4 * ((hh + 1) // 2) * ((ww + 15) // 16 + 1) + 4)
                    n_all += (fh * hip.row_stride(fw) + 31) // 32
                elif mode == 4:     # quads over the layer's own need map: four rows per group
                    caps.append(4 * ((hh + 3) // 4) * ((ww + 31) // 32 + 1) + 4)
                    n_all += (hh * hip.row_stride(ww) + 31) // 32
                elif mode == 0:
                    caps.append(hh * hip.row_stride(ww) // 32 + 2)
                    n_all += (hh * hip.row_stride(ww) + 31) // 32
                else:
                    caps.append((hh * hip.row_stride(ww) + bn - 1) // bn)
                    n_all += (hh * hip.row_stride(ww) + bn - 1) // bn
            staging_cap = max(caps)
            cap = sum(caps) + group * len(act)
            out = self._persist(("list", s, layer, mode, bn, group, shapes), lambda: torch.zeros(max(cap, 1), dtype=torch.int32, device=dev))
            stg = self._persist(("staging", s, layer, mode, bn, group, shapes),
                                lambda: torch.zeros(len(act) * staging_cap, dtype=torch.int32, device=dev))
            v = arr[s]
            v.layer, v.mode, v.bn, v.group = self.layer_index[layer], mode, bn, group
            v.pair_layer = self.layer_index[pair_layer] if pair_layer is not None else 0
            v.out, v.cap, v.staging, v.staging_cap = out.data_ptr(), cap, stg.data_ptr(), staging_cap
            self.specs.append({"out": out, "n_all": n_all, "group": group, "caps": caps, "layer": layer})
            self.outputs.append(out)
        self.lists_arr = arr
        L.n_lists, L.lists = len(specs), C.cast(arr, C.POINTER(hip.ViewList))
        L.summary = self.summary.data_ptr()
        nbytes = hip.lib.sm_view_lists_ws_bytes(C.byref(L))
        if nbytes == 0:
            raise RuntimeError("sm_view_lists_ws_bytes rejected the view's list table")
        self.ws = self._persist(("lists_ws", nbytes), lambda: torch.empty(nbytes + 256, dtype=torch.uint8, device=dev))
        L.ws, L.ws_bytes = self.ws.data_ptr(), self.ws.numel()

    # ------------------------------------------------------------------------------------------------------------
    def launch_masks(self, mask_u8, ag, adeg, rounded, other, interp_w):
        d = self.masks_desc
        d.mask, d.angle_degrees = mask_u8.data_ptr(), adeg.data_ptr()
        d.angle_guidance = ag.data_ptr() if (ag is not None and self.eng.cfg.use_angle_weight) else None
        if self.eng.cfg.use_depth_scaling:
            d.rounded, d.other, d.interp_w = rounded.data_ptr(), other.data_ptr(), interp_w.data_ptr()
        else:
            d.rounded = d.other = d.interp_w = None
        hip.check(hip.lib.sm_view_masks(C.byref(d), hip.stream()), "sm_view_masks")

    def launch_lists(self):
        if self.lists_desc is not None:
            hip.check(hip.lib.sm_view_lists(C.byref(self.lists_desc), hip.stream()), "sm_view_lists")

    def read_back(self):
        """Enqueue the one device-to-host copy of a view change; -> event that marks its completion."""
        self.summary_host.copy_(self.summary, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def parse_summary(self):
        """(view_tiles {job key: (list tensor, live fraction)}, mask sum per level) from the host copy of the summary."""
        host = self.summary_host
        n_specs = len(self.specs)
        ints = host[:9 * n_specs].view(n_specs, 9).tolist() if n_specs else []
        msums = host[9 * n_specs:9 * n_specs + len(self.level_hw)].view(torch.float32).tolist()
        tiles = None
        if self.lists_desc is not None:
            tiles = TileLists()
            tiles.quads = frozenset(j[0] for j in self.jobs if j[2] in (3, 4))
            for key, layer, mode, bn, group, pair_layer in self.jobs:
                s = self.job_spec[(layer, mode, bn, group, pair_layer)]
                sp, row = self.specs[s], ints[s]
                for g, cap in enumerate(sp["caps"]):
                    if row[1 + g] > cap:
                        raise RuntimeError(f"active list of {key}: {row[1 + g]} entries on level {g} exceed the buffer's {cap}")
                live = sum(row[1:1 + len(self.active)])
                tiles[key] = (sp["out"][:row[0]], live / max(sp["n_all"], 1))
        return tiles, msums

"""Every environment switch of the package in ONE table: name -> (default, kind, what it does).

kind:
  mode        selects between product paths that are all tested (arithmetic, communicator, backend);
  tuning      a product default with a measured alternative - the value here is what ships, the note says what the
              alternative measured;
  experiment  off by default, kept because a profile or DESIGN.md section 9 cites it (negative results stay runnable);
  diagnostic  prints / times, never changes results.

``tests/test_config_table.py`` keeps the table complete: every ``STYLEMESH_*`` / ``SM_*`` name read anywhere in the package,
bench.py or the C sources must be listed here, and every name listed here must still be read somewhere."""

SWITCHES = {
    # ---- arithmetic / kernels
    "STYLEMESH_CONV_MODE": ("split2", "mode", "conv arithmetic: split2 = fp16 x 2 operands, 3 products (default); "
                            "f32 = v_mfma_f32_32x32x2_f32 everywhere (bench.py's f32_mode leg, the parity tests' twin)"),
    "STYLEMESH_GRAM_MODE": ("(follows CONV_MODE)", "mode", "the same choice for the Gram forward / backward kernels"),
    "STYLEMESH_FUSE_POOL_BWD": ("1", "tuning", "max-pool backward taken by the data-gradient conv below the pool from argmax codes "
                                "(+4.5 % on c3, round 2); 0 = pool-backward kernels"),
    "STYLEMESH_FUSE_POOL_FWD": ("1", "tuning", "max-pool forward in the epilogue of the conv below it (+2.5 % on c3, round 3); 0 = pool kernels"),
    "STYLEMESH_FUSE_GRAM_BWD": ("r11,r21", "tuning", "style layers whose Gram backward rides in a data-gradient conv's epilogue "
                                "(+2.8 % together, round 3); 0 = none"),
    "STYLEMESH_SEGMENT_LISTS": ("1", "tuning", "active lists of 32-position segments (+10 % on c3, round 3); 0 = whole 128-position tiles"),
    "STYLEMESH_SEGMENT_STARTS": ("free", "tuning", "segment starts on the 4-position grid (+2.5 %, round 3); grid = aligned to 32"),
    "STYLEMESH_RESIDENT": ("1", "tuning", "round 5: the 64-output-channel conv launches (conv1_2 forward / data gradient, conv2_1's data "
                           "gradient) take lists of vertical segment QUADS and the resident-input kernel (SM_LIST_QUADS); 0 = the "
                           "ring kernel on 64 x 256 tiles"),
    "STYLEMESH_VIEW_CACHE_GB": ("0", "experiment", "round 5: HBM budget (GB) of the RESIDENT VIEWS - the per-view state (level maps, layer masks, "
                                "content targets, active lists, sorted scatter plan, touch flags) of a view stays on the device after its "
                                "first visit and is copied back at the next one instead of being recomputed (one rank, grouped view "
                                "path). +7.5 % when the view changes every step and views recur (bench.py's `resident_views` leg), "
                                "neutral on the reference's index_repeat-20 schedules (profiles/r05/resident_views.txt): opt-in"),
    "SM_CONV_SPLIT_PENALTY": ("3", "tuning", "(C library) cost of a K-split tail's second pass in tile-chunks, in the split-count "
                              "model (c2 +0.6 % at 2-4, -5 % at 8: profiles/r04/split_penalty_ab.txt)"),
    "STYLEMESH_VALIDATE_LISTS": ("0", "diagnostic", "1 = every quad list is copied to the host and checked against the preconditions of "
                                 "SM_LIST_QUADS before its launch (ops.check_quad_list: a sync per launch, debugging only)"),
    "SM_ADAM_DENSE_WALK": ("0", "diagnostic", "(C library) 1 = the flagged update walks every tile of the arena and asks each chunk's flag "
                           "(rounds 2-5) instead of compacting a span's flags first (adam_sparse_kernel, round 6; same bits of p, m, v)"),
    "SM_GRAM_TARGET_BLOCKS": ("(library default)", "experiment", "(C library) position-range count of the grouped Gram forward"),
    # ---- step structure
    "STYLEMESH_SIDE_STREAMS": ("1", "tuning", "loss branches of the non-deepest layers + the early half of the update on side streams "
                               "(+3 % on c3); 0 = one stream, the un-fused launch sequence; inline = the side-stream launch sequence (fused Gram "
                               "epilogues) issued on ONE stream - what the PMC passes profile"),
    "STYLEMESH_SIDE_STYLE": ("r11", "tuning", "style layers whose branch forks early (beside the deep convs)"),
    "STYLEMESH_EARLY_STYLE_AT": ("r31", "tuning", "conv output after which the early style branches fork (sweep: profiles/r03/c3_fork_point_sweep.txt)"),
    "STYLEMESH_EARLY_UPDATE_AT": ("r31", "tuning", "conv output after which the early half of the split update forks; head = before sampling"),
    "STYLEMESH_SPLIT_UPDATE": ("1", "tuning", "update of the chunks the view cannot reach beside the forward pass, closing update over "
                               "the view's own chunks only; 0 = one update"),
    "STYLEMESH_OVERLAP_MIN_PIXELS": ("400000", "tuning", "pixels over the active levels from which side streams pay (c2's 87 k do not)"),
    "STYLEMESH_SIDE_CUS": ("0", "experiment", "confine the side streams to N compute units (profiles/r03/side_stream_cu_subset_sweep_c3.txt: no gain)"),
    "STYLEMESH_MAIN_PRIORITY": ("high", "tuning", "trunk on a high-priority HIP stream (+2-4 %); normal = the caller's stream"),
    "STYLEMESH_STEP_PROGRAM": ("1", "mode", "small steps recorded once and replayed with one library call; 0 = always eager; "
                               "verify = record every step and compare it with the program"),
    "STYLEMESH_CONTENT_GRAPH": ("1", "tuning", "content-target VGG pass of a view replayed as one captured hipGraph; 0 = eager launches"),
    # ---- per-view work
    "STYLEMESH_FAST_VIEW": ("1", "mode", "per-view constants through the two grouped C entry points (sm_view_masks / sm_view_lists); "
                            "0 = the call-per-layer path the equality test compares against"),
    "STYLEMESH_PREPARE_AHEAD": ("1", "tuning", "next view's constants prepared one view ahead on a side stream (+2-3 %)"),
    # ---- multi-GPU
    "STYLEMESH_DIST_BACKEND": ("nccl", "mode", "process-group backend of bench.py / the launcher (gloo: functional runs with all ranks on one GPU)"),
    "STYLEMESH_COMM": ("(rccl over nccl groups)", "mode", "rccl = the product's own communicator (fails loudly), torch = torch.distributed"),
    "STYLEMESH_PIPELINE_EXCHANGE": ("0", "mode", "gradient exchange in pieces overlapped with the update: 1 / 0 = always / never; "
                                    "auto = from STYLEMESH_PIPELINE_MIN_MB of flagged chunks on (opt-in until timed over RCCL)"),
    "STYLEMESH_DEFERRED_EXCHANGE": ("0", "mode", "N > 1 with an owner-aware reducer: only chunks two or more ranks' views touch are exchanged "
                                    "before the update; single-owner chunks are updated by their owner at once and travel in the "
                                    "background (round 6; bit-identical over gloo, not timed over RCCL: opt-in)"),
    "STYLEMESH_PIPELINE_MIN_MB": ("32", "tuning", "threshold of the auto policy above"),
    "STYLEMESH_LAUNCHED_BY": ("(set by the launcher)", "diagnostic", "who started the ranks, echoed in the bench line"),
    # ---- host
    "STYLEMESH_HOST_THREADS": ("(from the cgroup quota)", "tuning", "cap of the training process's intra-op pool (runtime/hostcpu.py)"),
    "STYLEMESH_CPU_THREADS": ("(best of 32 / 64)", "tuning", "threads of bench.py's cpu_baseline leg"),
    # ---- diagnostics
    "STYLEMESH_SETVIEW_TIMING": ("0", "diagnostic", "host / GPU time of set_view's phases"),
    "STYLEMESH_TRAINER_TIMING": ("0", "diagnostic", "host seconds per phase of MiniTrainer's loop"),
    "STYLEMESH_MAIN_TIMING": ("0", "diagnostic", "start-up phases of the CLI"),
}


def describe() -> str:
    """The table as text (``python -m stylemesh_amd.runtime.config``)."""
    rows = [f"{k:34s} {v[0]:26s} {v[1]:11s} {v[2]}" for k, v in SWITCHES.items()]
    return "\n".join(rows)


if __name__ == "__main__":
    print(describe())

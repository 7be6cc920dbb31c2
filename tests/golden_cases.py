"""Constants shared by the golden generator's cases and the tests that replay them (keep in sync with
tests/golden/make_goldens.py: FLAGSETS, LOSS_WEIGHTS, STYLE_WEIGHTS, TEX, STYLE_HW, STYLE_SEED, VGG_SEED)."""
VGG_SEED = 7
STYLE_SEED = 43
STYLE_HW = (300, 270)
TEX = 64
LOSS_WEIGHTS = {"content": 7e1, "style": 1e-4, "tex_reg": 5e3}
STYLE_WEIGHTS = [1000., 1000., 10., 10., 1000.]
FLAGSETS = {
    "only2d": dict(hier=True, mode="single", gram="current", thr=3000, angle=False, depth=False, nlev=1),
    "with_angle": dict(hier=True, mode="multi", gram="current", thr=30, angle=True, depth=False, nlev=1),
    "with_angle_and_depth": dict(hier=True, mode="multi", gram="current", thr=30, angle=True, depth=True, nlev=2),
    "dip_average": dict(hier=True, mode="single", gram="average", thr=3000, angle=False, depth=False, nlev=1),
    "flat_single": dict(hier=False, mode="single", gram="current", thr=60, angle=True, depth=True, nlev=2),
}
SMALL_VIEW_HW = (40, 56)
SMALL_LEVEL_HW = [(40, 56), (64, 88)]
SMALL_ROOM = (6.0, 4.5, 2.8)
MULTIVIEW_SEEDS = [3, 4, 6, 8]

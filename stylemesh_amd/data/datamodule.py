"""Single-scene DataModule over synthetic views, with the reference DataModule's split / sampler semantics
(data/abstract_dataset.py:434-495): sequential split (first ``int(train_split * n)`` views train, rest val),
``RepeatingSampler`` with ``index_repeat``, batch size 1; for R ranks, rank r takes train views r, r + R, ...
The real ScanNet / Matterport directory loaders are the next scope row."""
from __future__ import annotations

from . import synthetic as S


class SyntheticSceneDataModule:
    split_modes = ["sequential"]
    sampler_modes = ["repeat", "sequential"]

    def __init__(self, n_views=8, view_hw=S.SCANNET_VIEW_HW, level_hw=None, min_pyramid_depth=0.25, split=(0.8, 0.2),
                 index_repeat=1, sampler_mode="repeat", room_size=(12.0, 9.0, 3.0), seed=0, rank=0, world_size=1,
                 use_depth_in_mask=True, prefetch=0):
        self.prefetch = prefetch
        self.n_views, self.view_hw = n_views, tuple(view_hw)
        self.level_hw = list(level_hw) if level_hw else [tuple(view_hw)]
        self.min_pyramid_depth, self.split, self.index_repeat = min_pyramid_depth, split, index_repeat
        self.sampler_mode, self.room, self.seed = sampler_mode, S.BoxRoom(room_size), seed
        self.rank, self.world_size, self.use_depth_in_mask = rank, world_size, use_depth_in_mask
        self._cache = {}

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        n_train = int(self.split[0] * self.n_views)
        self.train_indices = list(range(n_train))
        self.val_indices = list(range(n_train, self.n_views))

    def _view(self, i):
        if i not in self._cache:
            self._cache[i] = S.make_view(self.seed + i, view_hw=self.view_hw, level_hw=self.level_hw,
                                         level_heights=[h for h, _ in self.level_hw],
                                         min_pyramid_depth=self.min_pyramid_depth, room=self.room,
                                         use_depth_in_mask=self.use_depth_in_mask)
        return self._cache[i]

    def train_dataloader(self):
        from ..runtime.distributed import scheduled_batches   # equal step counts + lock-step view changes on every rank
        return scheduled_batches(self._view, self.train_indices, self.rank, self.world_size, self.index_repeat,
                                 repeat=self.sampler_mode == "repeat", prefetch=self.prefetch)

    def val_dataloader(self):
        return (self._view(i) for i in self.val_indices) if self.val_indices else None

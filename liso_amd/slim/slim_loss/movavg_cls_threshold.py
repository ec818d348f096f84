"""MovingAverageThreshold.  Mirror of liso/slim/slim_loss/movavg_cls_threshold.py:9-157 (same buffers -> same
state_dict keys: start_value, update_weight, [moving_counter, still_counter], bias_counter, moving_average_importance)."""
from typing import Tuple

import torch
import torch.distributed as dist
from torch import nn


class MovingAverageThreshold(nn.Module):
    def __init__(self, num_train_samples: int, num_moving: int, num_still=None, resolution: int = 100000,
                 start_value: float = 0.5, value_range: Tuple[float, float] = (0.0, 1.0), *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.value_range = (value_range[0], value_range[1] - value_range[0])
        self.resolution, self.num_moving, self.num_still = resolution, num_moving, num_still
        self.register_buffer("start_value", torch.tensor(start_value, dtype=torch.float))
        self.total = num_moving + (num_still if num_still is not None else 0)
        assert num_train_samples > 0, num_train_samples
        update_weight = 1.0 / min(2.0 * self.total, 5_000.0 * self.total / num_train_samples)
        self.register_buffer("update_weight", torch.tensor(update_weight, dtype=torch.double))
        if num_still is not None:
            self.register_buffer("moving_counter", torch.tensor(self.num_moving, dtype=torch.long))
            self.register_buffer("still_counter", torch.tensor(self.num_still, dtype=torch.long))
        self.register_buffer("bias_counter", torch.zeros((), dtype=torch.double))
        self.register_buffer("moving_average_importance", torch.zeros((self.resolution,), dtype=torch.float))
        # (extension) while this is a list, update() appends its per-rank histogram increment and count instead of reducing and
        # applying them: a trainer that replays the step from a hipGraph -- which cannot hold a collective -- reduces all increments
        # of the step in ONE all-reduce behind the replay and applies them in order (apply_deferred).  Within a step the threshold
        # is read once, before the first update (slim.py), so deferring the updates to the end of the step changes nothing.
        self._defer = None

    def value(self):
        """reference :56-60; the `if bias_counter > 0` branch is a torch.where (no device->host sync)"""
        return torch.where(self.bias_counter > 0.0, self._compute_optimal_score_threshold(), self.start_value)

    def _compute_bin_idxs(self, scores):
        idxs = ((scores - self.value_range[0]) * self.resolution / self.value_range[1]).to(torch.int)
        return torch.clamp(idxs, max=self.resolution - 1)

    def _compute_improvements(self, epes_stat_flow, epes_dyn_flow, moving_mask):
        if self.num_still is None:
            assert moving_mask is None
            return epes_stat_flow - epes_dyn_flow
        w = 1.0 / torch.where(moving_mask, self.moving_counter, self.still_counter).to(torch.float)
        return (epes_stat_flow - epes_dyn_flow) * w

    def _compute_optimal_score_threshold(self):
        z = torch.zeros((1,), dtype=self.moving_average_importance.dtype, device=self.moving_average_importance.device)
        improv = torch.cat([z, torch.cumsum(self.moving_average_importance, 0)], dim=0)
        is_min = torch.min(improv) == improv  # mean index of the minima, without nonzero()
        avg_idx = (torch.arange(improv.numel(), device=improv.device, dtype=torch.float) * is_min).sum() / is_min.sum()
        return self.value_range[0] + avg_idx * self.value_range[1] / self.resolution

    def _update_values(self, cur_value, cur_weight):
        w = (1.0 - self.update_weight) ** cur_weight
        self.moving_average_importance *= w.to(torch.float)
        self.moving_average_importance += (1.0 - w.to(torch.float)) * cur_value
        self.bias_counter *= w
        self.bias_counter += 1.0 - w

    def apply_deferred(self, items):
        """the updates recorded while `_defer` was a list: one SUM all-reduce of all histogram increments / counts over the ranks,
        then the moving-average updates in the recorded order (= what the immediate path does update by update)"""
        if not items:
            return
        cur = torch.stack([c for c, _ in items])
        cnt = torch.stack([n for _, n in items])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(cur)
            dist.all_reduce(cnt)
        for k in range(cur.shape[0]):
            self._update_values(cur[k], cnt[k])

    def update(self, epes_stat_flow, epes_dyn_flow, moving_mask, dynamicness_scores, training, valid_mask=None,
               compute_value=True):
        """reference :118-157.  Extensions: `valid_mask` -- rows to ignore (instead of the caller compacting the arrays
        with boolean indexing); `compute_value=False` skips evaluating the returned threshold."""
        assert isinstance(training, bool)
        if training:
            e_s, e_d, sc = epes_stat_flow.detach(), epes_dyn_flow.detach(), dynamicness_scores.detach()
            imp = self._compute_improvements(e_s, e_d, moving_mask)
            bins = self._compute_bin_idxs(sc).to(torch.long)
            count = e_s.numel()
            if valid_mask is not None:
                imp = torch.where(valid_mask, imp, 0.0)
                bins = torch.where(valid_mask, bins, 0).clamp(min=0)
                count = valid_mask.sum()
            cur = torch.zeros((self.resolution,), dtype=imp.dtype, device=imp.device).scatter_add_(0, bins, imp)
            if self._defer is not None:
                assert self.num_still is None, "deferred updates: unsupervised mode only (no moving / still counters)"
                self._defer.append((cur, torch.as_tensor(count, dtype=torch.long, device=imp.device).reshape(())))
                return self.value() if compute_value else None
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # data parallel: every rank applies the histogram increment of the GLOBAL batch (400 KB all-reduce), so
                # the threshold buffers stay identical on all replicas
                count = torch.as_tensor(count, dtype=torch.long, device=imp.device).clone()
                dist.all_reduce(cur)
                dist.all_reduce(count)
            self._update_values(cur, count)
            if self.num_still is not None:
                mm = moving_mask if valid_mask is None else moving_mask & valid_mask
                sm = ~moving_mask if valid_mask is None else (~moving_mask) & valid_mask
                n_mov, n_still = torch.count_nonzero(mm), torch.count_nonzero(sm)
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    # the counters weight the next step's improvements (_compute_improvements): like the histogram they
                    # must count the GLOBAL batch or the threshold buffers drift apart across replicas
                    both = torch.stack([n_mov, n_still])
                    dist.all_reduce(both)
                    n_mov, n_still = both[0], both[1]
                self.moving_counter += n_mov
                self.still_counter += n_still
        return self.value() if compute_value else None

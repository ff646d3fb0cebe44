"""IoU matcher and fg/bg sampler.

maskrcnn_benchmark/modeling/matcher.py:5-112 and
maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:5-68.
"""
import torch


class Matcher:
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        if match_quality_matrix.numel() == 0:
            if match_quality_matrix.shape[0] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        matched_vals, matches = match_quality_matrix.max(dim=0)
        all_matches = matches.clone() if self.allow_low_quality_matches else None
        below = matched_vals < self.low_threshold
        between = (matched_vals >= self.low_threshold) & (matched_vals < self.high_threshold)
        matches = torch.where(below, torch.full_like(matches, self.BELOW_LOW_THRESHOLD), matches)
        matches = torch.where(between, torch.full_like(matches, self.BETWEEN_THRESHOLDS), matches)
        if self.allow_low_quality_matches:
            # every prediction that ties a gt's best IoU keeps its original argmax (matcher.py:83-112)
            best_per_gt = match_quality_matrix.max(dim=1).values
            tied = (match_quality_matrix == best_per_gt[:, None]).any(dim=0)
            matches = torch.where(tied, all_matches, matches)
        return matches


class BalancedPositiveNegativeSampler:
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self._calls = 0

    def sample_device(self, labels, generator=None):
        """One image on the device (``_C.sample_fg_bg``): -> (selected [B] ascending indices, zero padded; positive_slots
        [B]: where the positives sit inside ``selected``; counts [2] int32 device tensor = selected, positives).  The
        subsets are uniformly random like the reference's randperm()[:k]; the key stream is seeded from torch's seed
        (or the given generator) and a per-sampler call counter, so runs with the same seeds repeat."""
        from .. import _C
        base = generator.initial_seed() if generator is not None else torch.initial_seed()
        self._calls += 1
        seed = (base * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        return _C.sample_fg_bg(labels, self.batch_size_per_image,
                               int(self.batch_size_per_image * self.positive_fraction), seed)

    def __call__(self, matched_idxs, generator=None):
        pos_idx, neg_idx = [], []
        for m in matched_idxs:
            positive = torch.nonzero(m >= 1).squeeze(1)
            negative = torch.nonzero(m == 0).squeeze(1)
            num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
            perm1 = torch.randperm(positive.numel(), device=positive.device, generator=generator)[:num_pos]
            perm2 = torch.randperm(negative.numel(), device=negative.device, generator=generator)[:num_neg]
            pos_mask = torch.zeros_like(m, dtype=torch.bool)
            neg_mask = torch.zeros_like(m, dtype=torch.bool)
            pos_mask[positive[perm1]] = True
            neg_mask[negative[perm2]] = True
            pos_idx.append(pos_mask)
            neg_idx.append(neg_mask)
        return pos_idx, neg_idx

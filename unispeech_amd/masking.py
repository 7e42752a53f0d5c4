"""Host-side span masking and relative-position bucketing (integer / boolean logic of the hot path).

Both stay on the host by design: the reference draws masks from the *global numpy
RNG* (src/fairseq/data/data_utils.py:393-517 == WavLM/WavLM.py:35-159) and "bit-exact mask indices" therefore
means: same numpy calls, same order, same arguments.  This module restates that procedure; tests/ pin it against
masks produced by the reference itself under the same seeds (tests/golden/).

RNG call order reproduced here (A.3 of SURVEY.md): one rand() for the batch-level count; per row [rand() if a
padding mask is given], the length draw for non-static selections, choice(sz - min_len, n, replace=False) (or the
recursive no-overlap placement); after all rows one choice(idc, min_len, replace=False) per row that exceeds the
batch minimum.
"""
from typing import Optional, Tuple

import numpy as np
import torch


def _span_lengths(kind, n, mask_length, mask_other):
    if kind == "static":
        return np.full(n, mask_length)
    if kind == "uniform":
        return np.random.randint(mask_other, mask_length * 2 + 1, size=n)
    if kind == "normal":
        draws = np.random.normal(mask_length, mask_other, size=n)
        return [max(1, int(round(v))) for v in draws]
    if kind == "poisson":
        draws = np.random.poisson(mask_length, size=n)
        return [int(round(v)) for v in draws]
    raise Exception("unknown mask selection " + kind)


def _place_without_overlap(sz, lengths, min_space):
    """largest spans first; each span is dropped uniformly into a free interval chosen with probability
    proportional to its length, then the interval is split around it"""
    chosen = []
    free = [(0, sz)]
    shortest = min(lengths)
    for span in sorted(lengths, reverse=True):
        room = np.fromiter((e - s if e - s >= span + min_space else 0 for s, e in free), np.int64)
        total = np.sum(room)
        if total == 0:
            break
        pick = np.random.choice(len(free), p=room / np.sum(room))
        s, e = free.pop(pick)
        start = np.random.randint(s, e - span)
        chosen.extend(start + o for o in range(span))
        if start - s - min_space >= shortest:
            free.append((s, start - min_space + 1))
        if e - start - shortest - min_space > shortest:
            free.append((start + span + min_space, e))
    return np.asarray(chosen)


def compute_mask_indices(shape: Tuple[int, int], padding_mask: Optional[torch.Tensor], mask_prob: float,
                         mask_length: int, mask_type: str = "static", mask_other: float = 0.0, min_masks: int = 0,
                         no_overlap: bool = False, min_space: int = 0) -> np.ndarray:
    """bool [B, T] span mask; every row ends up with the same number of masked frames (batch minimum)."""
    bsz, all_sz = shape
    mask = np.full((bsz, all_sz), False)
    batch_count = max(min_masks, int(mask_prob * all_sz / float(mask_length) + np.random.rand()))
    # (host time: this function runs on the launch thread once per step -- 7.5 ms at 32 x 749 frames in its first form, more
    # than half of the thread's time per step.  Same numpy calls in the same order; the span expansion and the per-row padding
    # counts are vectorised: 1.x ms)
    pad_counts = None
    if padding_mask is not None:
        pm = padding_mask.cpu() if torch.is_tensor(padding_mask) else torch.as_tensor(np.asarray(padding_mask))
        pad_counts = pm.reshape(bsz, -1).long().sum(dim=1).tolist()

    per_row = []
    for b in range(bsz):
        if pad_counts is not None:
            sz = all_sz - pad_counts[b]
            count = max(min_masks, int(mask_prob * sz / float(mask_length) + np.random.rand()))
        else:
            sz, count = all_sz, batch_count
        lengths = _span_lengths(mask_type, count, mask_length, mask_other)
        static = mask_type == "static"
        if (int(mask_length) * count if static else sum(lengths)) == 0:
            lengths[0] = min(mask_length, sz - 1)
            static = False
        if no_overlap:
            idc = _place_without_overlap(sz, lengths, min_space)
        else:
            shortest = int(mask_length) if static else min(lengths)
            if sz - shortest <= count:
                shortest = sz - count - 1
            starts = np.random.choice(sz - shortest, count, replace=False)
            if static:  # every span has the same length: the reference's nested comprehension as one outer sum
                idc = (starts[:, None] + np.arange(int(mask_length))[None, :]).reshape(-1)
            else:
                idc = np.asarray([starts[j] + o for j in range(len(starts)) for o in range(lengths[j])])
        flags = np.zeros(sz, dtype=bool)          # np.unique(idc[idc < sz]) without the sort: the distinct indices, ascending
        idc = np.asarray(idc, dtype=np.int64)
        flags[idc[idc < sz]] = True
        per_row.append(np.flatnonzero(flags))

    keep = min(len(r) for r in per_row)
    for b, idc in enumerate(per_row):
        if len(idc) > keep:
            idc = np.random.choice(idc, keep, replace=False)
        mask[b, idc] = True
    return mask


_BUCKET_CACHE = {}


def relative_position_buckets(T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """int32 [2T-1]: bucket of relative position d = j - i, indexed by d + T - 1.

    Same arithmetic, in the same order and dtypes, as MultiheadAttention._relative_positions_bucket with
    bidirectional=True (WavLM/modules.py:417-442): float32 log, python-float divisor, truncation to int64.  The
    reference evaluates it on the full [T, T] grid on the CPU every forward; the grid is Toeplitz, so the 2T-1
    distinct values are computed once per (T, num_buckets, max_distance) and cached.
    """
    key = (T, num_buckets, max_distance)
    hit = _BUCKET_CACHE.get(key)
    if hit is not None:
        return hit
    import math

    rel = torch.arange(-(T - 1), T, dtype=torch.long)
    half = num_buckets // 2
    out = (rel > 0).to(torch.long) * half
    n = torch.abs(rel)
    exact = half // 2
    small = n < exact
    large = exact + (torch.log(n.float() / exact) / math.log(max_distance / exact) * (half - exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, half - 1))
    out = out + torch.where(small, n, large)
    res = out.to(torch.int32)
    _BUCKET_CACHE[key] = res
    return res

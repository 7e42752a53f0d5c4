"""Batch collation with utterance / noise mixing for the MI355X path.

`UtteranceMixingCollater.collater(samples)` mirrors `UtteranceMixingDataset.collater`
(src/fairseq/data/audio/utterance_mixing_dataset.py:323-371) and, with mixing_prob = 0, `HubertDataset.collater`
(src/fairseq/data/audio/hubert_dataset.py:280-347): same sample dicts in ({"id", "source", "label_list", "boundary"}),
same batch dict out (id, net_input{source, padding_mask, boundary}, target_list / target_lengths_list / ntokens_list or
the single_target forms, task).

Division of labour:
  host   -- every random draw, from the global numpy stream in the reference's order (crop offsets, which rows are
            mixed, partners, span lengths / positions, SNRs), the label crop / padding (integer work on a few hundred
            labels), and packing the cropped waveforms into ONE pinned staging buffer;
  device -- one asynchronous H2D copy of that buffer and one kernel launch (wavlm_mix_utterances) doing what the
            reference does with per-sample numpy / torch loops on the CPU: the energy ratios (two full-row reductions
            per mix), the scaled span additions in the reference's in-place row order, the optional per-row
            normalisation, and the bf16 cast of the waveform the Trainer would do afterwards.
At 11 k audio-seconds/s per GPU (32 x 15 s every 43 ms) the reference's loop -- ~8 passes over a 240 000-sample row per
mixed utterance in numpy -- cannot feed one GPU from one worker; here the host touches each sample twice (the memcpy into staging and, when mixing is
on, the all-zero test of the staged row that decides whether the reference would draw an SNR).

There is no CPU mixing path in this module: `device=None` returns the plan and the un-mixed batch (used by the CPU
tests together with the oracle); with a device the HIP library does the arithmetic or the call raises.
"""
from typing import List, Optional

import numpy as np
import torch

from . import functional as F


def collate_tokens(values, pad_idx, left_pad=False):
    """data_utils.collate_tokens (src/fairseq/data/data_utils.py:33-73) for 1-D tensors, no eos handling"""
    size = max(v.size(0) for v in values) if len(values) else 0
    res = values[0].new(len(values), size).fill_(pad_idx) if len(values) else torch.zeros(0, 0, dtype=torch.long)
    for i, v in enumerate(values):
        dst = res[i][size - len(v):] if left_pad else res[i][:len(v)]
        dst.copy_(v)
    return res


class UtteranceMixingCollater:
    def __init__(self, sample_rate=16000, label_rates=(50,), pad_list=(1,), max_sample_size=None, pad_audio=False,
                 normalize=False, random_crop=False, single_target=False, multitask=False, mixing_max_len=-1,
                 mixing_prob=0.2, mixing_num=1, mixing_noise=False, mixing_noise_prob=0.0, mixing_noise_num=1,
                 noise_list=None, noise_loader=None, device=None, out_dtype=torch.float32):
        """Arguments as UtteranceMixingDataset.__init__ (utterance_mixing_dataset.py:84-118) where they concern collation.
        noise_list: entries with a "loc" field as the reference's noise manifest; noise_loader(entry) -> 1-D float32 numpy
        array already scaled to [-1, 1] (the reference reads int16 from an h5 file and divides by 32767)."""
        self.sample_rate = sample_rate
        self.label_rates = list(label_rates)
        self.pad_list = list(pad_list)
        self.num_labels = len(self.label_rates)
        self.max_sample_size = max_sample_size if max_sample_size is not None else 2 ** 62
        self.pad_audio, self.normalize, self.random_crop = pad_audio, normalize, random_crop
        self.single_target, self.multitask = single_target, multitask
        self.mixing_max_len, self.mixing_prob, self.mixing_num = mixing_max_len, mixing_prob, mixing_num
        self.mixing_noise, self.mixing_noise_prob, self.mixing_noise_num = mixing_noise, mixing_noise_prob, mixing_noise_num
        self.noise_list, self.noise_loader = noise_list, noise_loader
        self.device = torch.device(device) if device is not None else None
        self.out_dtype = out_dtype
        self._stages = [None, None, None]  # rotating pinned staging buffers: [tensor, event of the last H2D copy out of it]
        self._stage_i = 0

    # ---------------------------------------------------------------------------------------------- host: audio
    def crop_to_max_size(self, n, target_size):
        """(start, end) of the crop; utterance_mixing_dataset.py:310-321"""
        diff = n - target_size
        if diff <= 0:
            return 0, n
        start, end = 0, target_size
        if self.random_crop:
            start = np.random.randint(0, diff + 1)
            end = n - diff + start
        return start, end

    def _staging(self, B, T):
        """next pinned staging buffer of the rotation.  The H2D copy out of a staging buffer is asynchronous and the
        training loop never synchronises (defer_logging), so the host can be a batch or two ahead of the copy engine: a
        buffer is handed out again only after the event recorded behind its last copy has completed (`_staged_copy`).
        PyTorch's pinned allocator guards a block against free + reuse only, not against host writes into a live tensor."""
        n = B * T
        self._stage_i = (self._stage_i + 1) % len(self._stages)
        slot = self._stages[self._stage_i]
        if slot is None or slot[0].numel() < n:
            buf = torch.empty(max(n, 1), dtype=torch.float32)
            if self.device is not None and self.device.type == "cuda":
                buf = buf.pin_memory()
            slot = self._stages[self._stage_i] = [buf, None]
        if slot[1] is not None:
            slot[1].synchronize()  # the copy engine has read the previous batch out of this buffer
            slot[1] = None
        return slot[0][:n].view(B, T)

    def _staged_copy(self, staged):
        """asynchronous H2D copy of the current staging buffer + the event that guards the buffer's reuse"""
        dev_t = staged.to(self.device, non_blocking=True)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._stages[self._stage_i][1] = ev
        return dev_t

    def collater_audio(self, audios, audio_size):
        """utterance_mixing_dataset.py:440-462: crop / zero-pad into [B, audio_size]; returns (staging tensor, padding
        mask, audio starts, all-zero flag of every collated row)"""
        B = len(audios)
        out = self._staging(B, audio_size)
        padding_mask = torch.zeros(B, audio_size, dtype=torch.bool)
        starts = [0] * B
        zero = [False] * B
        for i, a in enumerate(audios):
            diff = len(a) - audio_size
            if diff == 0:
                out[i].copy_(a)
            elif diff < 0:
                assert self.pad_audio
                out[i, :len(a)].copy_(a)
                out[i, len(a):].zero_()
                padding_mask[i, diff:] = True
            else:
                s, e = self.crop_to_max_size(len(a), audio_size)
                out[i].copy_(a[s:e])
                starts[i] = s
            if self.mixing_prob > 0:
                # power of the COLLATED row (after crop / zero-pad), as mixing_collated_audios measures it
                # (utterance_mixing_dataset.py:420-421): a crop can be silent although the utterance is not
                zero[i] = not bool(out[i].any())
        return out, padding_mask, starts, zero

    # ------------------------------------------------------------------------------------------- host: mix plan
    def draw_mixing_plan(self, B, T, row_is_zero=None):
        """The random part of mixing_collated_audios (utterance_mixing_dataset.py:373-438): same numpy calls in the same
        order.  Returns (ops int32 [n, 8], op_begin int32 [B + 1], noise float32 [total] or None).
        op = (row, kind 0 = batch row / 1 = noise segment, src row | noise offset, c_start, s_start, c_len, src length,
        float32 bits of 10 ** (snr / 10)).
        The reference draws the SNR only when the partner's power is non-zero.  That is data dependent, but decidable on
        the host without touching the waveform again: a row's power is zero iff the row is all zeros, and mixing never
        changes that (a zero row has ref_pow = 0 -> scale 0; a non-zero row stays non-zero) -- `row_is_zero[r]` of the
        COLLATED rows before mixing (after crop / zero-pad: `collater_audio` tests the staged row) is enough."""
        mixing_max_len = T // 2 if self.mixing_max_len < 0 else T // self.mixing_max_len
        mixing_max_len = T if mixing_max_len > T else mixing_max_len
        zero = [False] * B if row_is_zero is None else list(row_is_zero)
        ops, begin, noise_chunks, noise_off = [], [0], [], 0
        for i in range(B):
            if np.random.random() < self.mixing_prob:
                if self.mixing_noise and np.random.random() < self.mixing_noise_prob:
                    choices = np.random.choice(self.noise_list, self.mixing_noise_num)
                    for c in choices:
                        nz = np.ascontiguousarray(self.noise_loader(c), dtype=np.float32)
                        gain = 1.0
                        if np.any(nz):
                            snr = np.random.uniform(-5, 20)
                            gain = 10 ** (snr / 10)
                        c_len = np.random.randint(0, mixing_max_len + 1)
                        c_len = min(c_len, nz.shape[0])
                        c_end = np.random.randint(c_len, nz.shape[0] + 1)
                        s_end = np.random.randint(c_len, T + 1)
                        ops.append((i, 1, noise_off, c_end - c_len, s_end - c_len, c_len, nz.shape[0], gain))
                        noise_chunks.append(nz)
                        noise_off += nz.shape[0]
                else:
                    choices = np.random.choice(range(B), self.mixing_num, replace=True)
                    for c in choices:
                        c = int(c)
                        c_len = np.random.randint(0, mixing_max_len + 1)
                        c_end = np.random.randint(c_len, T + 1)
                        s_end = np.random.randint(c_len, T + 1)
                        gain = 1.0
                        if not zero[c]:
                            snr = np.random.uniform(-5, 5)
                            gain = 10 ** (snr / 10)
                        ops.append((i, 0, c, c_end - c_len, s_end - c_len, c_len, T, gain))
            begin.append(len(ops))
        arr = np.zeros((len(ops), 8), dtype=np.int32)
        for k, o in enumerate(ops):
            arr[k, :7] = o[:7]
            arr[k, 7] = np.float32(o[7]).view(np.int32)
        noise = np.concatenate(noise_chunks) if noise_chunks else None
        return arr, np.asarray(begin, dtype=np.int32), noise

    # --------------------------------------------------------------------------------------------- host: labels
    def collater_frm_label(self, targets, audio_size, audio_starts, label_rate, pad):
        """utterance_mixing_dataset.py:464-486"""
        assert label_rate > 0
        s2f = label_rate / self.sample_rate
        frm_starts = [int(round(s * s2f)) for s in audio_starts]
        frm_size = int(round(audio_size * s2f))
        if not self.pad_audio:
            rem_size = [len(t) - s for t, s in zip(targets, frm_starts)]
            frm_size = min(frm_size, *rem_size)
        targets = [t[s: s + frm_size] for t, s in zip(targets, frm_starts)]
        lengths = torch.LongTensor([len(t) for t in targets])
        ntokens = lengths.sum().item()
        return collate_tokens(targets, pad_idx=pad, left_pad=False), lengths, ntokens

    def collater_seq_label(self, targets, pad):
        lengths = torch.LongTensor([len(t) for t in targets])
        return collate_tokens(targets, pad_idx=pad, left_pad=False), lengths, lengths.sum().item()

    def collater_label(self, targets_by_label, audio_size, audio_starts):
        tl, ll, nl = [], [], []
        for targets, label_rate, pad in zip(targets_by_label, self.label_rates, self.pad_list):
            if label_rate == -1:
                t, l, n = self.collater_seq_label(targets, pad)
            else:
                t, l, n = self.collater_frm_label(targets, audio_size, audio_starts, label_rate, pad)
            tl.append(t); ll.append(l); nl.append(n)
        return tl, ll, nl

    # ------------------------------------------------------------------------------------------------- collater
    def collater(self, samples):
        samples = [s for s in samples if s["source"] is not None]
        if len(samples) == 0:
            return {}
        audios = [s["source"] for s in samples]
        sizes = [len(a) for a in audios]
        bnds = [s.get("boundary", []) for s in samples]
        audio_size = min(max(sizes), self.max_sample_size) if self.pad_audio else min(min(sizes), self.max_sample_size)
        staged, padding_mask, starts, zero = self.collater_audio(audios, audio_size)
        B, T = staged.shape
        plan = None
        if self.mixing_prob > 0:
            plan = self.draw_mixing_plan(B, T, zero)
        targets_by_label = [[s["label_list"][i] for s in samples] for i in range(self.num_labels)]
        targets_list, lengths_list, ntokens_list = self.collater_label(targets_by_label, audio_size, starts)

        if self.device is None:
            source = staged.clone()
            pm_dev = padding_mask
        else:
            src_dev = self._staged_copy(staged)
            pm_dev = F.h2d(padding_mask, self.device)
            if plan is None:
                source = src_dev if self.out_dtype == torch.float32 else src_dev.to(self.out_dtype)
            else:
                from . import ops
                ops_np, begin_np, noise_np = plan
                source = ops.mix_utterances(src_dev, F.h2d(ops_np.reshape(-1), self.device), ops_np.shape[0],
                                            F.h2d(begin_np, self.device),
                                            F.h2d(noise_np, self.device) if noise_np is not None else None,
                                            self.normalize, self.out_dtype)
        net_input = {"source": source, "padding_mask": pm_dev, "padding_mask_cpu": padding_mask, "boundary": bnds}
        batch = {"id": torch.LongTensor([s["id"] for s in samples]), "net_input": net_input}
        if self.single_target:
            batch["target_lengths"], batch["ntokens"], batch["target"] = lengths_list[0], ntokens_list[0], targets_list[0]
        else:
            batch["target_lengths_list"], batch["ntokens_list"], batch["target_list"] = lengths_list, ntokens_list, targets_list
        batch["task"] = "multitask" if self.multitask else "wavlm"
        if plan is not None:
            batch["mixing_plan"] = plan  # (ops, op_begin, noise): what the device kernel was given (tests / debugging)
        return batch

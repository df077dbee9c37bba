"""Batch assembly for the hot path: the reference's collate + SpecAugment, with the padded batch
built directly in device memory.

* `collate_fn_with_eos_bos(batch, device)`: otrans/data/loader.py:66-108 -- pad features with zeros, bool masks,
  targets `[BOS] tokens [EOS] PAD*`, `targets_length` counting the EOS.  Features may already live on the device (they are
  copied into ONE preallocated [B,Tmax,F] buffer; no per-utterance F.pad + cat).
* `spec_augment_ranges` + `spec_augment_batch`: otrans/data/augment.py:9-41.  The mask rectangles are drawn on the host
  with the reference's exact sequence of `np.random.uniform` / `random.randint` calls (so a seeded run produces the same
  masks), then applied to the whole batch by one kernel (otr_spec_mask) instead of per-utterance numpy slicing.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib as L

PAD, BOS, EOS = 0, 1, 1          # otrans/data/__init__.py:7-12


def collate_fn_with_eos_bos(batch, device=None):
    """batch: list of (utt_id, feat [T_i, F] float tensor, feat_len, target (list of ints), target_len)."""
    utt_ids = [d[0] for d in batch]
    features_length = [int(d[2]) for d in batch]
    targets_length = [int(d[4]) for d in batch]
    Tmax, Lmax = max(features_length), max(targets_length)
    device = torch.device(device) if device is not None else batch[0][1].device
    F_ = batch[0][1].shape[-1]
    feats = torch.zeros((len(batch), Tmax, F_), dtype=torch.float32, device=device)
    tg = torch.full((len(batch), Lmax + 2), PAD, dtype=torch.long)
    for i, (_, feat, flen, target, tlen) in enumerate(batch):
        feats[i, :flen].copy_(feat[:flen], non_blocking=True)
        tg[i, 0] = BOS
        tg[i, 1:1 + tlen] = torch.as_tensor(list(target)[:tlen], dtype=torch.long)
        tg[i, 1 + tlen] = EOS
    flen_t = torch.tensor(features_length, dtype=torch.int32)
    fmask = torch.arange(Tmax).unsqueeze(0) < flen_t.unsqueeze(1)
    tlen_t = torch.tensor(targets_length, dtype=torch.int32)
    tmask = torch.arange(Lmax + 2).unsqueeze(0) < (tlen_t + 2).unsqueeze(1)
    inputs = {'inputs': feats, 'inputs_length': flen_t.to(device), 'mask': fmask.to(device)}
    targets = {'targets': tg.to(device), 'targets_length': (tlen_t + 1).to(device), 'mask': tmask.to(device)}
    return utt_ids, inputs, targets


def spec_augment_ranges(tau, v, freq_mask_num=2, time_mask_num=2, freq_mask_rate=0.3, time_mask_rate=0.05,
                        max_mask_time_len=100):
    """The rectangles data/augment.py:9-41 would zero for a [tau, v] spectrogram, as rows {t0, t1, f0, f1}; consumes
    numpy's and `random`'s global generators in exactly the reference's order."""
    freq_para = int(v * freq_mask_rate)
    time_para = min(int(tau * time_mask_rate), max_mask_time_len)
    out = []
    for _ in range(freq_mask_num):
        f = int(np.random.uniform(low=0.0, high=freq_para))
        f0 = random.randint(0, v - f)
        out.append((0, tau, f0, f0 + f))
    for _ in range(time_mask_num):
        t = int(np.random.uniform(low=0.0, high=time_para))
        t0 = random.randint(0, tau - t)
        out.append((t0, t0 + t, 0, v))
    return out


def spec_augment_batch(inputs, lengths=None, **kw):
    """In-place SpecAugment of a padded device batch [B,T,F]: utterance b is treated as a [lengths[b], F] spectrogram
    (the reference augments each utterance before padding)."""
    if not inputs.is_cuda:
        raise L.OtransHipError('spec_augment_batch needs a CUDA/HIP tensor; there is no CPU fallback')
    B, T, F_ = inputs.shape
    lengths = [T] * B if lengths is None else [int(x) for x in lengths]
    rows = [spec_augment_ranges(lengths[b], F_, **kw) for b in range(B)]
    NR = len(rows[0])
    if NR == 0:
        return inputs
    r = torch.tensor(rows, dtype=torch.int32).reshape(B, NR, 4).to(inputs.device)
    x = inputs if inputs.is_contiguous() else inputs.contiguous()
    L.check(L.load().otr_spec_mask(C.c_void_p(x.data_ptr()), C.c_void_p(r.data_ptr()), B, NR, T, F_,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_spec_mask')
    if x is not inputs:
        inputs.copy_(x)
    return inputs

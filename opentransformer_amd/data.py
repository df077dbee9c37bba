"""Batch assembly for the hot path: the reference's collate + SpecAugment, with the padded batch
built directly in device memory.

* `collate_fn_with_eos_bos(batch, device)`: otrans/data/loader.py:66-108 -- pad features with zeros, bool masks,
  targets `[BOS] tokens [EOS] PAD*`, `targets_length` counting the EOS.  Features may already live on the device (they are
  copied into ONE preallocated [B,Tmax,F] buffer; no per-utterance F.pad + cat).
* `BySequenceLengthSampler`: otrans/data/bucket.py:14-170 -- length-bucketed batches (same batches as the reference under
  the same `random` seed), which bound the padding of those device batches.
* `spec_augment_ranges` + `spec_augment_batch`: otrans/data/augment.py:9-41.  The mask rectangles are drawn on the host
  with the reference's exact sequence of `np.random.uniform` / `random.randint` calls (so a seeded run produces the same
  masks), then applied to the whole batch by one kernel (otr_spec_mask) instead of per-utterance numpy slicing.
"""
import ctypes as C
import random

import numpy as np
import torch
import torch.utils.data

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


class BySequenceLengthSampler(torch.utils.data.Sampler):
    """Length-bucketed batch sampler with the reference's semantics (otrans/data/bucket.py:14-170), so that a seeded
    `random` yields the very same batches: utterances go to the bucket whose (lower, upper] frame range holds them, each
    bucket is shuffled and cut into batches -- of `bucket_batch_size[min(id, last)]` utterances, or (the default,
    `audo_set_batch_size`, spelled as in the reference) greedily up to `max_frames_one_batch` frames -- and the batches of
    all buckets are shuffled together at the start of every epoch.  Padding per batch is bounded by the bucket width,
    which is what keeps the padded [B, Tmax, F] device batches of `collate_fn_with_eos_bos` dense.

    Reference quirks kept on purpose (they change which batches come out):
      * `short_first` is accepted and ignored (bucket.py:28 overwrites it with False);
      * in frame-budget mode an utterance longer than the budget that comes first in its bucket emits an EMPTY batch
        before it (bucket.py:139-142), and `drop_last` drops the last partial batch of every bucket;
      * the per-bucket lists are shuffled in place, so `shuffle_batch_in_bucket()` (called between epochs) reshuffles the
        previous epoch's order rather than the dataset order.
    Bucket lookup is one `np.searchsorted` over all lengths instead of a numpy round trip per utterance."""
    INT32_MAX = int(np.iinfo(np.int32).max)

    def __init__(self, dataset, bucket_boundaries, bucket_batch_size=[], rm_the_long_sents=True,
                 audo_set_batch_size=True, max_frames_one_batch=20000, drop_last=False, short_first=False):
        assert isinstance(bucket_boundaries, list) and isinstance(bucket_batch_size, list)
        self.index_length_pair = list(dataset.index_length_pair())
        self.bucket_boundaries = bucket_boundaries
        self.bucket_batch_size = bucket_batch_size
        self.rm_the_long_sents = rm_the_long_sents
        self.audo_set_batch_size = audo_set_batch_size
        self.max_frames_one_batch = max_frames_one_batch
        self.drop_last = drop_last
        self.short_first = False
        if audo_set_batch_size:
            assert max_frames_one_batch > 0
        self.max_length = bucket_boundaries[-1] if rm_the_long_sents else self.INT32_MAX
        self.num_of_buckets = len(bucket_boundaries) + (0 if rm_the_long_sents else 1)
        self.buckets = {}
        for b in range(self.num_of_buckets):
            lo = 0 if b == 0 else bucket_boundaries[b - 1]
            hi = self.INT32_MAX if b == len(bucket_boundaries) else bucket_boundaries[b]
            size = 0 if audo_set_batch_size else bucket_batch_size[min(b, len(bucket_batch_size) - 1)]
            self.buckets[str(b)] = {'index_length_pair': [], 'batch_size': size, 'boundary': [lo, hi]}
        self.batch_list = []
        self.put_data_pair_into_buckets()
        self.split_the_bucket_into_batch()

    def element_to_bucket_id(self, seq_length):
        if seq_length <= 0 or seq_length > self.INT32_MAX:
            raise ValueError('no bucket holds a sequence of length %r' % (seq_length,))
        return int(np.searchsorted(np.asarray(self.bucket_boundaries), seq_length, side='left'))

    def put_data_pair_into_buckets(self):
        pairs = self.index_length_pair
        if not pairs:
            return
        lengths = np.asarray([p[1] for p in pairs])
        keep = lengths <= self.max_length if self.rm_the_long_sents else np.ones(len(pairs), bool)
        if np.any(lengths[keep] <= 0):
            raise ValueError('no bucket holds a sequence of length <= 0')
        ids = np.searchsorted(np.asarray(self.bucket_boundaries), lengths, side='left')
        for (index, length), b, k in zip(pairs, ids, keep):
            if k:
                self.buckets[str(int(b))]['index_length_pair'].append((index, length))
        self.removed = int(len(pairs) - keep.sum())

    def split_the_bucket_into_batch(self):
        self.batch_list = []
        for bucket_id, bucket in self.buckets.items():
            pairs = bucket['index_length_pair']
            if not pairs:
                continue
            random.shuffle(pairs)
            if self.audo_set_batch_size:
                self.batch_list.extend(self.generate_batches_based_length(bucket_id, pairs))
            else:
                self.batch_list.extend(self.generate_batches(bucket_id, pairs, bucket['batch_size']))

    def generate_batches(self, bucket_id, pairs, batch_size):
        n = len(pairs)
        nb = n // batch_size if self.drop_last else -(-n // batch_size)
        return [(int(bucket_id), [i for i, _ in pairs[s * batch_size:min((s + 1) * batch_size, n)]]) for s in range(nb)]

    def generate_batches_based_length(self, bucket_id, pairs):
        out, cur, frames = [], [], 0
        for index, length in pairs:
            if frames + length > self.max_frames_one_batch:
                out.append((int(bucket_id), cur))
                cur, frames = [], 0
            cur.append(index)
            frames += length
        if cur and not self.drop_last:
            out.append((int(bucket_id), cur))
        return out

    def shuffle_batch_in_bucket(self):
        if not self.short_first:
            self.split_the_bucket_into_batch()

    def padding_waste(self):
        """fraction of padded frames over all current batches (what bucketing is for): 1 - sum(len) / sum(B * Tmax)"""
        length = dict(self.index_length_pair)
        real = padded = 0
        for _, idx in self.batch_list:
            if idx:
                ls = [length[i] for i in idx]
                real += sum(ls)
                padded += len(ls) * max(ls)
        return 1.0 - real / max(padded, 1)

    def __iter__(self):
        if not self.short_first:
            random.shuffle(self.batch_list)
        for _, index_list in self.batch_list:
            yield index_list

    def __len__(self):
        return len(self.batch_list)

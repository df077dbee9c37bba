"""Batch assembly against the reference's collate_fn_with_eos_bos / spec_augment
(tests/golden/data_collate.npz, oracle/make_golden.py:golden_data)."""
import random

import numpy as np
import pytest
import torch


def toy_batch():
    """(utt_id, feat [T,F], feat_len, target, target_len) like the reference datasets yield."""
    rng = np.random.default_rng(5)
    out = []
    for i, (T, L) in enumerate([(37, 4), (52, 7), (20, 1), (45, 5)]):
        feat = torch.from_numpy(rng.standard_normal((T, 40)).astype(np.float32))
        tgt = [int(v) for v in rng.integers(3, 50, L)]
        out.append(('utt%d' % i, feat, T, tgt, L))
    return out


class ToyLengths:
    """what the sampler needs from a dataset: (index, number of frames) pairs; a few utterances beyond the last boundary"""
    def index_length_pair(self):
        rng = np.random.default_rng(9)
        return [(i, int(v)) for i, v in enumerate(rng.integers(30, 1300, 400))]


BUCKET_CASES = {
    'frames': dict(bucket_boundaries=[200, 400, 600, 800, 1000], max_frames_one_batch=6000),
    'frames_keep_long_drop_last': dict(bucket_boundaries=[300, 600, 900], rm_the_long_sents=False, max_frames_one_batch=5000,
                                       drop_last=True, short_first=True),
    'fixed': dict(bucket_boundaries=[250, 500, 750, 1000], bucket_batch_size=[16, 12, 8], audo_set_batch_size=False),
    'fixed_drop_last_keep_long': dict(bucket_boundaries=[400, 800], bucket_batch_size=[10, 6, 3], audo_set_batch_size=False,
                                      rm_the_long_sents=False, drop_last=True),
    'frames_budget_below_one_utterance': dict(bucket_boundaries=[500, 1000], max_frames_one_batch=700),
}


def flatten_batches(seq):
    vals = np.asarray([i for b in seq for i in b], dtype=np.int64)
    offs = np.cumsum([0] + [len(b) for b in seq]).astype(np.int64)
    return vals, offs


def test_bucket_sampler_matches_reference(golden):
    """same `random` seed -> the same batches as data/bucket.py, through construction, two epochs and a re-split"""
    from opentransformer_amd.data import BySequenceLengthSampler
    g = golden('data_bucket.npz')
    for name, kw in BUCKET_CASES.items():
        random.seed(21)
        s = BySequenceLengthSampler(ToyLengths(), **kw)
        seqs = [[b for _, b in s.batch_list], list(s), list(s)]
        s.shuffle_batch_in_bucket()
        seqs.append(list(s))
        for i, seq in enumerate(seqs):
            v, o = flatten_batches(seq)
            assert np.array_equal(o, g['%s_%d_o' % (name, i)]), (name, i)
            assert np.array_equal(v, g['%s_%d_v' % (name, i)]), (name, i)
        assert len(s) == len(seqs[-1])


def test_bucket_sampler_bounds_padding():
    from opentransformer_amd.data import BySequenceLengthSampler
    random.seed(3)
    ds = ToyLengths()
    bucketed = BySequenceLengthSampler(ds, [200, 400, 600, 800, 1000, 1300], max_frames_one_batch=8000)
    one = BySequenceLengthSampler(ds, [1300], max_frames_one_batch=8000)
    assert bucketed.padding_waste() < 0.15 < one.padding_waste()
    seen = sorted(i for b in bucketed for i in b)
    assert seen == list(range(400))                      # every utterance exactly once
    with pytest.raises(ValueError):
        bucketed.element_to_bucket_id(0)


def test_collate_matches_reference(golden):
    from opentransformer_amd.data import collate_fn_with_eos_bos
    g = golden('data_collate.npz')
    ids, inputs, targets = collate_fn_with_eos_bos(toy_batch(), device='cpu')
    assert ids == ['utt0', 'utt1', 'utt2', 'utt3']
    assert np.array_equal(inputs['inputs'].numpy(), g['inputs'])
    assert np.array_equal(inputs['inputs_length'].numpy(), g['inputs_length'])
    assert np.array_equal(inputs['mask'].numpy(), g['mask'])
    assert np.array_equal(targets['targets'].numpy(), g['targets'])
    assert np.array_equal(targets['targets_length'].numpy(), g['targets_length'])
    assert np.array_equal(targets['mask'].numpy(), g['targets_mask'])
    assert inputs['mask'].dtype == torch.bool and targets['targets'].dtype == torch.long


def test_spec_augment_ranges_match_reference(golden):
    """same seeds -> same rectangles as data/augment.py (applied here with numpy; the GPU test applies them with the kernel)"""
    from opentransformer_amd.data import spec_augment_ranges
    g = golden('data_collate.npz')
    np.random.seed(11)
    random.seed(12)
    for i, (_, feat, flen, _, _) in enumerate(toy_batch()):
        x = feat[:flen].numpy().copy()
        for t0, t1, f0, f1 in spec_augment_ranges(flen, x.shape[1], time_mask_rate=0.2):
            x[t0:t1, f0:f1] = 0
        assert np.array_equal(x, g['aug%d' % i]), i


@pytest.mark.gpu
def test_spec_augment_batch_on_device_matches_reference(golden):
    from opentransformer_amd.data import collate_fn_with_eos_bos, spec_augment_batch
    g = golden('data_collate.npz')
    batch = toy_batch()
    _, inputs, _ = collate_fn_with_eos_bos(batch, device='cuda')
    np.random.seed(11)
    random.seed(12)
    x = spec_augment_batch(inputs['inputs'], lengths=[b[2] for b in batch], time_mask_rate=0.2)
    for i, b in enumerate(batch):
        assert np.array_equal(x[i, :b[2]].cpu().numpy(), g['aug%d' % i]), i
        assert float(x[i, b[2]:].abs().sum()) == 0.0

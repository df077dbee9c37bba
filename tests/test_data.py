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

"""CPU: evaluation / checkpoint tooling against the reference's own average_parameters
(fixture made by oracle/make_golden.py) and hand-checked edit distances."""
import os

import numpy as np
import torch

from opentransformer_amd import tools


def make_checkpoints(expdir, n=5):
    """Deterministic toy checkpoints in the reference's layout ({'params','frontend','encoder','decoder'})."""
    for e in range(n):
        rng = np.random.default_rng(100 + e)
        w = torch.from_numpy(rng.standard_normal((4, 3)).astype(np.float32))
        state = {'params': {'epoch': e}, 'epochs': e, 'global_step': 10 * e,
                 'frontend': {'a.weight': w.clone(), 'a.bias': torch.full((4,), float(e))},
                 'encoder': {'b.weight': w * 2, 'b.bias': torch.arange(3, dtype=torch.float32) * e},
                 'decoder': {'embedding.weight': w + 1, 'output_layer.weight': w + 1}}
        torch.save(state, os.path.join(expdir, 'model.epoch.%d.pt' % e))


def test_average_parameters_matches_reference(tmp_path, golden):
    g = golden('tools_average.npz')
    make_checkpoints(str(tmp_path), 5)
    out = tools.average_parameters(str(tmp_path), N=3)
    assert os.path.basename(out) == 'model.average.last.3.pt'
    state = torch.load(out)
    assert state['params'] == {'epoch': 2} and state['epochs'] == 2          # settings of the first of the last N
    for name in g.files:
        part, key = name.split('/', 1)
        np.testing.assert_allclose(state[part][key].numpy(), g[name], rtol=1e-6, atol=1e-7)


def test_edit_distance_and_scoring():
    assert tools.edit_distance('kitten', 'sitting') == 3
    assert tools.edit_distance([], [1, 2]) == 2 and tools.edit_distance([1, 2, 3], [1, 2, 3]) == 0
    assert tools.edit_distance('a b c d'.split(), 'a x c'.split()) == 2
    r = tools.score_hypotheses(['a b c d', 'e f'], [['a x c', 'a b c'], ['e f', 'zzz']])
    assert r['total_tokens'] == 6 and r['false_tokens'] == 2 and r['top_n_false_tokens'] == 1
    assert abs(r['wer'] - 100 * 2 / 6) < 1e-9 and abs(r['topn_wer'] - 100 / 6) < 1e-9


def test_average_parameters_integer_buffers(tmp_path):
    """BatchNorm's num_batches_tracked (int64) makes the reference's div_ raise under current torch
    (`result type Float can't be cast to Long`); here integer buffers are floor-averaged instead."""
    for e in range(3):
        torch.save({'params': {}, 'encoder': {'bn.num_batches_tracked': torch.tensor(10 * e), 'w': torch.full((2,), float(e))}},
                   os.path.join(str(tmp_path), 'model.epoch.%d.pt' % e))
    state = torch.load(tools.average_parameters(str(tmp_path), N=3))
    assert int(state['encoder']['bn.num_batches_tracked']) == 10 and state['encoder']['w'].tolist() == [1.0, 1.0]

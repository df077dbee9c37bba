"""Host-side evaluation / checkpoint tooling of the reference: checkpoint averaging
(otrans/utils.py:46-102) and the WER/CER bookkeeping of eval.py:123-192 without the `editdistance` dependency.
Pure CPU logic: nothing here touches the GPU library."""
import glob
import os

import torch


def average_parameters(expdir, N=20):
    """Average the last N `model.epoch.<e>.pt` checkpoints of `expdir` (otrans/utils.py:46-102): every entry of the
    checkpoint dict except 'params' / 'epochs' / 'amp' / 'global_step' is a state_dict and is averaged key by key;
    the settings come from the first of the N checkpoints.  Writes model.average.last.<N>.pt and returns its path."""
    chkpts = glob.glob(os.path.join(expdir, 'model.epoch.*.pt'))
    assert len(chkpts) >= N
    last_n = sorted(chkpts, key=lambda x: int(x.split('.')[-2]))[-N:]
    new_state, sums = None, {}
    for path in last_n:
        state = torch.load(path, map_location='cpu')
        if new_state is None:
            new_state = state
        for key, sd in state.items():
            if key in ('params', 'epochs', 'amp', 'global_step'):
                continue
            acc = sums.setdefault(key, {})
            for k, p in sd.items():
                if k not in acc:
                    acc[k] = p.clone()          # clone: p may be a shared (tied) parameter
                else:
                    acc[k] += p
    for key, acc in sums.items():
        for k in acc:
            if acc[k].is_floating_point():
                acc[k].div_(N)
            else:                                # BatchNorm's int64 num_batches_tracked: the reference's div_ raises on
                acc[k] = acc[k] // N             # current torch; floor-average it instead
        new_state[key] = acc
    out = os.path.join(expdir, 'model.average.last.%d.pt' % N)
    torch.save(new_state, out)
    return out


def edit_distance(ref, hyp):
    """Levenshtein distance between two token sequences (what editdistance.eval returns, eval.py:168)."""
    ref, hyp = list(ref), list(hyp)
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def score_hypotheses(truths, nbest_preds):
    """eval.py:123-192: truths = list of reference strings, nbest_preds = list (per utterance) of n-best hypothesis
    strings (best first).  Returns WER (1-best) and top-n WER (oracle over the n-best) in percent plus the counts."""
    total_tokens = false_tokens = top_n_false = 0
    for truth, preds in zip(truths, nbest_preds):
        ref = truth.split()
        total_tokens += len(ref)
        best = None
        for i, pred in enumerate(preds):
            n_diff = edit_distance(ref, pred.split())
            if i == 0:
                false_tokens += n_diff
            best = n_diff if best is None else min(best, n_diff)
        top_n_false += best if best is not None else len(ref)
    return {'wer': false_tokens / max(total_tokens, 1) * 100, 'topn_wer': top_n_false / max(total_tokens, 1) * 100,
            'false_tokens': false_tokens, 'top_n_false_tokens': top_n_false, 'total_tokens': total_tokens}

"""CPU (-m "not gpu"): the C-ABI library builds, loads and exports exactly what include/otrans_hip.h
declares; argument validation works without a GPU (negative return, no launch); the product refuses
CPU tensors instead of falling back."""
import ctypes as C
import os
import re

import pytest
import torch

from opentransformer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'otrans_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(otr_[a-z0-9_]+)\s*\(', src)))


def test_header_binding_and_library_agree():
    lib = _lib.load()
    decl = declared_symbols()
    assert decl, 'no declarations parsed'
    assert sorted(_lib.SIGNATURES) == decl          # the ctypes stub binds exactly the header
    for name in decl:
        assert hasattr(lib, name), name             # and the .so exports every one of them
    # the ABI version: header constant == library answer == what the binding was written against (breaking changes of
    # descriptors / signatures must bump it, and _lib.load refuses a library that answers anything else)
    hdr = int(re.search(r'#define\s+OTR_ABI_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 'otrans_hip.h')).read()).group(1))
    assert lib.otr_version() == hdr == _lib.OTR_ABI_VERSION


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    d = _lib.LinearDesc(4, 4, 0, 0, 0, 0, 0, 4, 4, 4, 0, 0)       # K = 0
    assert lib.otr_linear_fwd(C.byref(d), None, None, None, None, None, 0, None) < 0
    assert b'linear' in lib.otr_last_error_string()
    a = _lib.AttnDesc(1, 1, 4, 4, 24, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0)   # head dim not built
    assert lib.otr_attention_fwd(C.byref(a), None, None, None, None, None, None, None) < 0
    ln = _lib.LnDesc(4, 6, 0, 1e-5, 0.0, 0)                          # d % 4 != 0
    assert lib.otr_add_layernorm_fwd(C.byref(ln), None, None, None, None, None, None, None, None, None, None, None) < 0


def test_optimizer_step_validates_before_it_launches():
    """every check of otr_optimizer_step sits in front of its first launch (a refused call must not advance the device
    state), and a caller that allocated the old 16-float state block is refused instead of being written out of bounds"""
    lib = _lib.load()
    al, odd = C.c_void_p(4096), C.c_void_p(4096 + 4)

    def call(param=al, grad=al, m=al, v=al, state_floats=_lib.OTR_OPT_STATE_FLOATS, lp=None, noise=0.0):
        return lib.otr_optimizer_step(param, grad, m, v, 1024, al, state_floats, lp, 1e-3, 0.9, 0.98, 1e-9, 0.0, 1.0, 5.0, 256.0, 100.0,
                                      1.0, 2.0, noise, None)
    assert call(state_floats=16) < 0 and b'OTR_OPT_STATE_FLOATS' in lib.otr_last_error_string()
    assert call(param=odd) < 0 and b'aligned' in lib.otr_last_error_string()
    assert call(m=odd) < 0 and call(grad=odd) < 0 and call(lp=C.c_void_p(4096 + 2)) < 0
    assert call(noise=-1.0) < 0


def test_argument_errors_of_the_round5_entries():
    """the entries added in round 5 validate before they launch (no GPU here): the one-launch loss, the token-view embedding, the
    step-start fill, the scaled cast, the positional encoding with the mask cast"""
    lib = _lib.load()
    al, odd = C.c_void_p(4096), C.c_void_p(4096 + 4)

    def loss(logits=al, ld=4240, ldt=16, L=15, R=480, V=4234, dl=al, ldd=4240, ticket=al, scratch=al, dt=_lib.OTR_F32):
        return lib.otr_label_smoothing_loss_fused(logits, ld, al, ldt, L, R, V, 0.1, 0, None, al, dl, dt, ldd, scratch, ticket, None)
    assert loss(ticket=None) < 0 and b'label_smoothing_loss_fused' in lib.otr_last_error_string()
    assert loss(ld=4234) < 0            # rows not 16-byte aligned (ld % 4 != 0): the three-kernel form serves those
    assert loss(logits=odd) < 0
    assert loss(R=481) < 0              # R % L != 0
    assert loss(ldt=14) < 0             # row stride of the target view shorter than L
    assert loss(V=9000, ld=9000, ldd=9000) < 0
    assert loss(R=480 * 32, L=15) < 0   # > 8192 rows
    assert loss(dt=7) < 0 and loss(dt=_lib.OTR_BF16, ldd=4236, ld=4236) < 0      # 16-bit gradient rows need ld % 8 == 0
    assert lib.otr_embed_posenc_fwd_ld(al, 14, al, al, None, 480, 15, 256, 4234, 16.0, None) < 0       # ld_tok < L
    assert lib.otr_embed_bwd_ld(al, 16, 15, None, None, 0, al, 480, 256, 4234, 16.0, None) < 0         # neither dy nor slabs
    assert lib.otr_embed_bwd_ld(al, 16, 15, None, None, 4, al, 480, 256, 4234, 16.0, None) < 0         # nslab without slabs
    assert lib.otr_zero_tick(odd, 1024, None, 0, None) < 0 and b'zero_tick' in lib.otr_last_error_string()
    assert lib.otr_zero_tick(None, 0, None, 0, None) == 0                                                # nothing to do
    assert lib.otr_scale_cast(odd, al, 1024, 16.0, None) < 0 and lib.otr_scale_cast(None, al, 8, 1.0, None) < 0
    assert lib.otr_posenc_mask_fwd(al, al, None, 64, 0, 256, 16.0, None, 0, 0, None, None) < 0             # T = 0 (r06: d % 4 != 0 is served again)
    assert lib.otr_posenc_mask_fwd(al, al, None, 64, 8, 256, 16.0, al, 8, 1, None, None) < 0             # mask_in without mask_out


def test_argument_errors_of_the_fused_and_grouped_entries():
    """the entries added for fusion / grouping / decoding validate before they launch (no GPU here)"""
    lib = _lib.load()
    one = C.c_void_p(16)                        # non-null, 16-byte "aligned" dummy: must never be dereferenced
    assert lib.otr_ffn_glu_fwd(None, 8, None, 8, None, None, None, 4, 64, 8, None) < 0
    assert b'ffn_glu_fwd' in lib.otr_last_error_string()
    rows = C.c_int32(7)
    assert lib.otr_ffn_glu_bwd(one, 1, 4, one, 4, one, 1, one, one, 1, C.byref(rows), 4, 64, 8, None) < 0   # ldy < d_model
    assert lib.otr_linear_wgrad_grouped(None, -1, 1, None, 0, None) < 0
    assert lib.otr_linear_wgrad_grouped(None, 0, 1, None, 0, None) == 0          # empty group: nothing to do
    assert lib.otr_linear_wgrad_grouped(None, 0, 7, None, 0, None) < 0           # bad compute type
    it = (_lib.ColsumItem * 1)()
    it[0].a, it[0].out, it[0].M, it[0].N, it[0].lda, it[0].dtype = 16, 16, 4, 8, 8, 5
    assert lib.otr_colsum_grouped(it, 1, None) < 0                               # bad dtype
    assert lib.otr_colsum_grouped(None, 0, None) == 0
    assert lib.otr_transpose_batched(one, one, one, 1, 1, 2, None) < 0           # in place
    assert lib.otr_spec_mask(one, one, 1, 65, 4, 4, None) < 0                    # too many rectangles
    assert lib.otr_decode_embed(None, 4, None, None, None, None, 2, 8, 10, 1.0, None) < 0
    assert lib.otr_decode_self_attention(one, one, one, one, one, one, 1, 2, 4, 256, 8, 1.0, None) < 0      # dk > 128
    assert lib.otr_beam_prune_cached(one, one, one, one, one, 8, 1, 2, 1, one, one, one, one, 4, one, one, one, one,
                                     None) < 0                                   # in/out buffers must differ
    hcode = lib.otr_half_type()
    ld0 = _lib.LinearDesc(8, 8, 8, hcode, hcode, 0, hcode, 8, 8, 8, 0, 0)
    assert lib.otr_linear_fwd_batched(C.byref(ld0), one, one, None, 4, 8, 8, 8, None) < 0                 # null output
    assert lib.otr_linear_fwd_batched(C.byref(ld0), one, one, one, 0, 8, 8, 8, None) < 0                  # no batches
    assert b'linear_fwd_batched' in lib.otr_last_error_string()
    assert lib.otr_dwconv_bwd_part(one, one, hcode, one, one, None, 2, 8, 16, 5, 2, None) < 0              # no partial buffer
    assert lib.otr_dwconv_bwd_partial_rows(7968) == 249                                                   # 32 rows per workgroup
    lnd = _lib.LnDesc(8, 16, 0, 1e-5, 0.0, 0)
    assert lib.otr_add_layernorm2_fwd(C.byref(lnd), one, one, one, one, None, one, None, one, None, one, one, one, one, one, None) < 0   # gamma2 missing
    assert lib.otr_add_layernorm2_bwd(C.byref(lnd), one, one, one, one, one, one, one, one, one, None, None, one, one, None, None) < 0   # no partial buffer
    assert b'add_layernorm2_bwd' in lib.otr_last_error_string()
    assert lib.otr_dwconv_fwd_part(one, hcode, one, None, one, None, 2, 8, 16, 5, 2, None) < 0                 # no partial buffer
    assert lib.otr_bn_swish_fwd_part(one, one, 0, one, one, None, None, one, one, hcode, 16, 16, 1e-5, 0.1, None) < 0   # no partial rows
    assert lib.otr_dwconv_fwd_partial_rows(7968) == 249
    ln0 = _lib.DecLn(None, 16, None, 0, None, None, None, None, 0.0, 1e-5, 0, None, None, None, None, None)
    assert lib.otr_dec_self_step(C.byref(ln0), 8, one, one, one, one, one, None, one, 4, one, None) < 0     # ancestor table missing
    assert b'dec_self_step' in lib.otr_last_error_string()
    assert lib.otr_dec_self_step(C.byref(ln0), 0, one, one, one, one, one, one, one, 4, one, None) < 0      # no rows
    assert lib.otr_dec_self_step(C.byref(ln0), 8, one, one, one, one, one, one, one, 0, one, None) < 0      # cache length
    assert lib.otr_add_layernorm_bwd_partial_rows(7968) == 996            # 8 rows per workgroup (r06: 8 waves x 1 row)
    assert lib.otr_act_fwd(one, one, 0, 64, 4, None) < 0                         # kind must be gelu / tanh / swish
    assert b'act_fwd' in lib.otr_last_error_string()
    assert lib.otr_act_bwd(one, None, one, 1, 64, 1, None) < 0                   # dy missing
    assert lib.otr_act_fwd(C.c_void_p(24), one, 0, 64, 1, None) < 0              # unaligned
    assert lib.otr_act_fwd(one, one, 1, 0, 3, None) == 0                         # nothing to do


def test_product_refuses_cpu_tensors():
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn
    model = ota.SpeechToText(syn.c1_model())
    inputs, targets = syn.synthetic_batch(2, 100, 80, 100, 5)
    with pytest.raises(_lib.OtransHipError):
        model(inputs, targets)


def test_state_dict_keys_match_reference_layout():
    """a reference checkpoint must load unchanged."""
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn
    from tests import helpers as H
    cfg = syn.c1_model(ctc_weight=0.3)
    model = ota.SpeechToText(cfg)
    want = H.empty_state(cfg)
    for part, mod in (('frontend', model.frontend), ('encoder', model.encoder), ('decoder', model.decoder),
                      ('ctc', model.assistor)):
        got = mod.state_dict()
        assert sorted(got) == sorted(want[part]), part
        for k in got:
            assert tuple(got[k].shape) == tuple(want[part][k].shape), (part, k)
    assert model.decoder.output_layer.weight is model.decoder.embedding.weight   # tied


def test_reference_trained_checkpoint_loads(golden):
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn
    g = golden('c1_decode.npz')
    model = ota.SpeechToText(syn.c1_model(ctc_weight=0.3))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w:')}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def test_wgrad256_eligibility_rule_is_host_logic():
    """otr_wgrad256_takes (no GPU needed): the 256-wide weight-gradient launch takes long-contraction 16-bit problems in
    16-byte aligned rows and nothing else; otr_linear_wgrad_grouped refuses a bias gradient on an item it would not take."""
    import ctypes as C
    from opentransformer_amd import _lib as L
    for kind, code in (('bf16', L.OTR_BF16), ('fp16', L.OTR_F16)):
        lib = L.load(kind)
        lib.otr_debug_set(6, 1)
        try:
            def item(M, N, K, ldy=None, ldx=None, dy_dtype=code, x_dtype=code, base=0x10000):
                it = L.WgradItem()
                it.dy, it.x, it.dw = base, base + 0x100000, base + 0x200000
                it.M, it.N, it.K = M, N, K
                it.ldy, it.ldx, it.ldw = ldy or N, ldx or K, K
                it.dy_dtype, it.x_dtype = dy_dtype, x_dtype
                it.dbias = None
                return it
            takes = lambda it: lib.otr_wgrad256_takes(C.byref(it), code)        # noqa: E731
            assert takes(item(7968, 768, 256)) == 1
            assert takes(item(7968, 256, 608)) == 1                      # ragged last tile of K
            assert takes(item(512, 4096, 256)) == 1                      # the decoder's rows
            assert takes(item(7968, 512, 256, ldy=3072)) == 1            # a slice of a wider matrix
            assert takes(item(128, 768, 256)) == 0                       # short contraction
            assert takes(item(7968, 100, 256)) == 0 and takes(item(7968, 256, 68)) == 0    # narrow / not whole 16-byte units
            assert takes(item(7968, 768, 256, ldy=772)) == 0             # rows not 16-byte aligned
            assert takes(item(7968, 768, 256, dy_dtype=L.OTR_F32)) == 0  # fp32 operand
            assert takes(item(7968, 768, 256, base=0x10004)) == 0        # misaligned base
            assert lib.otr_wgrad256_takes(C.byref(item(7968, 768, 256)), L.OTR_F32) == 0   # fp32 compute mode
            lib.otr_debug_set(6, 0)
            assert takes(item(7968, 768, 256)) == 0                      # switched off
        finally:
            lib.otr_debug_set(6, -1)


def _wgrad256_pieces(plan, starts, Ms):
    """tests' replay of wgrad256_kernel's work decoding (csrc/wgrad256.hip: slot <- block, pieces of a slot)"""
    mode, G, chunk, nfull, rem, parts, total, n = plan
    pieces = []
    prob_of = lambda pos: max(i for i in range(n) if starts[i] <= pos)      # noqa: E731
    for bid in range(G):
        gx, rounds = G >> 3, mode == 1
        if G & 7 == 0:
            slot = (bid & 7) * gx + ((bid >> 3) if rounds else gx - 1 - (bid >> 3))
        else:
            slot = bid if rounds else G - 1 - bid
        if rounds:
            R, rnd = chunk, 0
            while True:
                if rnd < nfull:
                    t, sb, P, sl, ns = rnd * G + slot, 0, R, 0, 1
                elif rnd == nfull and slot < rem * parts:
                    part = slot % parts
                    t, sb = nfull * G + slot // parts, R * part // parts
                    P, sl, ns = R * (part + 1) // parts - sb, part, parts
                else:
                    break
                rnd += 1
                pi = prob_of(t * R)
                pieces.append((pi, (t * R - starts[pi]) // R, sb, P, sl, ns))
        else:
            pos, c_end = slot * chunk, min(slot * chunk + chunk, total)
            while pos < c_end:
                pi = prob_of(pos)
                R = (Ms[pi] + 15) // 16
                rel = pos - starts[pi]
                tile = rel // R
                sb = rel - tile * R
                tb = starts[pi] + tile * R
                te = tb + R
                pe = min(te, c_end)
                first, last = tb // chunk, (te - 1) // chunk
                pieces.append((pi, tile, sb, pe - pos, last - slot, last - first + 1))
                pos = pe
    return pieces


def test_wgrad256_schedule_covers_every_slab_once():
    """otr_debug_wgrad256_plan (host only) + the kernel's decoding replayed here: for the rounds schedule (equal row counts) and
    the stream-K schedule (mixed row counts, forced), every (problem, tile, 16-row slab) is worked on exactly once, no piece is
    empty, and the pieces of a tile carry the slice indices 0 .. n-1 of the turnstile exactly once"""
    import ctypes as C
    from opentransformer_amd import _lib as L
    lib = L.load('bf16')

    def items_of(shapes):
        arr = (L.WgradItem * len(shapes))()
        for it, (M, N, K) in zip(arr, shapes):
            it.dy, it.x, it.dw = 0x10000, 0x20000, 0x30000
            it.M, it.N, it.K, it.ldy, it.ldx, it.ldw = M, N, K, N, K, K
            it.dy_dtype = it.x_dtype = L.OTR_BF16
            it.dbias = None
        return arr
    layer = [(7968, 768, 256), (7968, 256, 256), (7968, 4096, 256), (7968, 256, 2048)]
    cases = [(layer * 12 + [(7968, 512, 256)] * 6 + [(7968, 256, 608)], 0),      # the training step: rounds, 1 full round + halves
             (layer * 12, -248), (layer * 12, 96), (layer, 0), (layer, 3),
             ([(512, 768, 256), (512, 256, 256), (512, 4096, 256), (512, 256, 2048)] * 6, 0),           # the decoder group
             ([(1032, 256, 256), (2048, 512, 256), (1544, 256, 768)], 0), ([(1032, 384, 136), (7968, 256, 608)], 7)]
    for shapes, cap in cases:
        out = (C.c_int32 * (8 + len(shapes)))()
        assert lib.otr_debug_wgrad256_plan(items_of(shapes), len(shapes), cap, out) == 0
        plan, starts = list(out[:8]), list(out[8:])
        Ms = [s[0] for s in shapes]
        assert plan[1] >= 1 and (plan[0] == 1) == (len(set(Ms)) == 1 and cap >= 0 and sum(-(-n // 256) * -(-k // 256) for _, n, k in shapes) >= 8)
        seen, slices = {}, {}
        for pi, tile, sb, P, sl, ns in _wgrad256_pieces(plan, starts, Ms):
            R = (Ms[pi] + 15) // 16
            tiles = -(-shapes[pi][1] // 256) * -(-shapes[pi][2] // 256)
            assert 0 <= tile < tiles and P >= 1 and 0 <= sb and sb + P <= R and 0 <= sl < ns, (shapes[pi], tile, sb, P, sl, ns)
            for sidx in range(sb, sb + P):
                assert (pi, tile, sidx) not in seen
                seen[(pi, tile, sidx)] = 1
            slices.setdefault((pi, tile), []).append((sl, ns))
        want = sum(-(-n // 256) * -(-k // 256) * ((m + 15) // 16) for m, n, k in shapes)
        assert len(seen) == want == plan[6], (len(seen), want, plan)
        for key, lst in slices.items():
            ns = lst[0][1]
            assert all(n == ns for _, n in lst) and sorted(s for s, _ in lst) == list(range(ns)), (key, lst)


def test_conv2_dgrad_plan_and_index_arithmetic():
    """otr_debug_conv2_dgrad_plan (host only) + the implicit conv2 input-gradient kernel's pixel / tap arithmetic replayed here
    (csrc/conv.hip conv2_dgrad_class): the four parity classes with their workgroup ranges cover every act1 pixel exactly once,
    and the taps a class uses -- (kh, kw) -> g2 pixel (t2, f2), dropped when out of range -- are exactly the (t2, f2, kh, kw) for
    which the forward convolution (3x3, stride 2, pad (0, 1): frontend/conv.py:63) reads that pixel."""
    import ctypes as C
    import numpy as np
    from opentransformer_amd import _lib as L
    from opentransformer_amd.ops import conv_geometry
    lib = L.load('bf16')
    for (B, T, Fd, C1, C2) in [(32, 1000, 80, 64, 128), (3, 97, 40, 64, 128), (2, 200, 83, 32, 64), (1, 7, 3, 64, 128), (5, 331, 80, 128, 128)]:
        T1, F1, T2, F2 = conv_geometry(T, Fd)
        desc = L.ConvDesc(B, T, Fd, C1, C2, T1, F1, T2, F2, L.OTR_BF16, L.OTR_BF16, L.OTR_BF16)
        out = (C.c_int32 * 10)()
        assert lib.otr_debug_conv2_dgrad_plan(C.byref(desc), out) == 0
        served, wg0, tiles = out[0], list(out[1:6]), list(out[6:10])
        assert served == int((C1, C2) in ((64, 128), (32, 64)))
        assert wg0[0] == 0 and wg0[4] <= 512
        cover = np.zeros((B, T1, F1), np.int32)
        small = B * T1 * F1 <= 40000                      # the tap relation is checked exhaustively on the small shapes
        for c in range(4):
            pt, pf = c >> 1, c & 1
            nT, nF = (T1 // 2 if pt else (T1 + 1) // 2), (F1 // 2 if pf else (F1 + 1) // 2)
            Mc = B * nT * nF
            assert tiles[c] == -(-Mc // 256)
            nwg = wg0[c + 1] - wg0[c]
            assert (nwg >= 1) == (tiles[c] > 0) and nwg <= max(tiles[c], 0)
            # workgroup w of the class walks tiles w, w + nwg, ...: every tile once
            walked = sorted(t for w in range(nwg) for t in range(w, tiles[c], nwg))
            assert walked == list(range(tiles[c]))
            m = np.arange(Mc)
            bi, j = m // nF, m % nF
            b, i = bi // nT, bi % nT
            t1, f1 = 2 * i + pt, 2 * j + pf
            np.add.at(cover, (b, t1, f1), 1)
            nkw = 2 if pf else 1
            ntaps = (1 if pt else 2) * nkw
            if not small:
                continue
            for mm in range(Mc):
                got = set()
                for tt in range(ntaps):
                    a, b2 = tt // nkw, tt % nkw
                    kh, kw = (1 if pt else 2 * a), (2 * b2 if pf else 1)
                    t2, f2 = (i[mm] if pt else i[mm] - a), (j[mm] + 1 - b2 if pf else j[mm])
                    if 0 <= t2 < T2 and 0 <= f2 < F2:
                        got.add((int(t2), int(f2), kh, kw))
                want = {(t2, f2, kh, kw) for kh in range(3) for kw in range(3)
                        for t2 in [(t1[mm] - kh) // 2] if (t1[mm] - kh) % 2 == 0 and 0 <= t2 < T2
                        for f2 in [(f1[mm] + 1 - kw) // 2] if (f1[mm] + 1 - kw) % 2 == 0 and 0 <= f2 < F2}
                assert got == want, (c, mm, got, want)
        assert int(cover.min()) == 1 and int(cover.max()) == 1


def test_split_ffn_and_attention_grids_cover_every_unit_once():
    """Host-only views of two launch geometries whose decoding happens inside the kernels: under both workgroup mappings of the
    split FFN kernels every (row block, slice) pair is taken by exactly one workgroup, the four workgroups of a row block are
    within 32 ids of each other (co-resident on an idle GPU) and -- mapping 1 -- an XCD (id mod 8) sees one slice only; the
    XCD-aware attention grid gives every (block, head, utterance) exactly one workgroup and puts the blocks of a (head, utterance)
    on ids that are equal modulo 8."""
    import numpy as np
    from opentransformer_amd import _lib as L
    lib = L.load('bf16')
    for M in (128, 2048, 7968, 8000, 100000):
        nblk = (M + 127) // 128
        for wg_map in (0, 1):
            grid = lib.otr_debug_ffn_split_map(M, wg_map, None, 0)
            out = np.zeros(2 * grid, dtype=np.int32)
            assert lib.otr_debug_ffn_split_map(M, wg_map, out.ctypes.data_as(C.c_void_p), grid) == grid
            rb, sl = out[0::2], out[1::2]
            live = rb < nblk
            pairs = sorted(zip(rb[live].tolist(), sl[live].tolist()))
            assert pairs == [(r, s) for r in range(nblk) for s in range(4)], (M, wg_map)
            ids = np.arange(grid)[live]
            for r in (0, nblk // 2, nblk - 1):
                mine = ids[rb[live] == r]
                assert mine.max() - mine.min() < 32
            if wg_map == 1:
                for x in range(8):
                    assert len(set(sl[live][ids % 8 == x].tolist())) <= 1
    for nx, H, B in ((4, 4, 32), (8, 4, 32), (1, 4, 32), (2, 6, 5), (3, 1, 1)):
        grid = lib.otr_debug_attention_grid(nx, H, B, None, 0)
        out = np.zeros(3 * grid, dtype=np.int32)
        assert lib.otr_debug_attention_grid(nx, H, B, out.ctypes.data_as(C.c_void_p), grid) == grid
        trip = out.reshape(-1, 3)
        live = trip[:, 0] >= 0
        got = sorted(map(tuple, trip[live].tolist()))
        assert got == sorted((x, h, b) for b in range(B) for h in range(H) for x in range(nx)), (nx, H, B)
        ids = np.arange(grid)[live]
        for h, b in ((0, 0), (H - 1, B - 1)):
            sel = (trip[live][:, 1] == h) & (trip[live][:, 2] == b)
            assert len(set((ids[sel] % 8).tolist())) == 1


def test_gemm_tile_order_is_a_permutation():
    """csrc/gemm_kernel.h gemm_tile_of (the XCD-aware position -> tile map of the tile GEMM), emulated: a bijection of
    [0, tiles_m * tiles_n) for every shape, and positions that are equal modulo 8 walk the column tiles of the same row blocks."""
    def tile_of(t, tiles_m, tiles_n):
        full = (tiles_m >> 3) * 8 * tiles_n
        if t < full:
            idx = t >> 3
            g = idx // tiles_n
            return g * 8 + (t & 7), idx - g * tiles_n
        r = t - full
        q = r // tiles_n
        return (tiles_m & ~7) + q, r - q * tiles_n

    for tiles_m in (1, 7, 8, 9, 16, 63, 125, 189):
        for tiles_n in (1, 2, 3, 6, 67):
            seen = {tile_of(t, tiles_m, tiles_n) for t in range(tiles_m * tiles_n)}
            assert seen == {(m, n) for m in range(tiles_m) for n in range(tiles_n)}, (tiles_m, tiles_n)
    # 125 row blocks x 6 column tiles (7968 x 384 on 64-wide tiles): the six column tiles of a row block sit on ONE position class
    cls = {}
    for t in range(125 * 6):
        m, n = tile_of(t, 125, 6)
        cls.setdefault(m, set()).add(t % 8)
    assert all(len(v) == 1 for m, v in cls.items() if m < 120)

"""Index-math emulator of csrc/wgrad256.hip (numpy, CPU).  Mirrors the kernel's lane formulas one to one: the DMA image of
a slab, the transposing read (ds_read_b64_tr_b16 modelled as: within a 16-lane group, lane p receives element p%4 of the
8-byte pieces addressed by lanes 4j + p//4, j = 0..3), the 32x32x16 MFMA operand / accumulator layouts and the epilogue
transposition.  Checks dw == dy^T x for one 256 x 256 tile.  Run: python tools/emu/wgrad256_emu.py"""
import numpy as np

SLAB_ROWS, SLAB_BYTES, RING = 16, 16384, 8


def tr_read(lds16, addr_bytes):
    """lds16: LDS as uint16 array; addr_bytes[64]: per-lane byte address.  Returns [64,4] uint16."""
    out = np.zeros((64, 4), np.uint16)
    for lane in range(64):
        g, p = lane >> 4, lane & 15
        for j in range(4):
            src_lane = 16 * g + 4 * j + (p >> 2)
            a = addr_bytes[src_lane]
            assert a % 8 == 0
            out[lane, j] = lds16[a // 2 + (p & 3)]
    return out


def main(M=40, ldy=512, ldx=256, n0=256, k0=0, seed=0):
    rng = np.random.default_rng(seed)
    dy = rng.integers(-3, 4, size=(M, ldy)).astype(np.int32)     # small ints: exact in any arithmetic
    x = rng.integers(-3, 4, size=(M, ldx)).astype(np.int32)
    dy16 = dy.astype(np.int16).view(np.uint16)
    x16 = x.astype(np.int16).view(np.uint16)
    P = (M + SLAB_ROWS - 1) // SLAB_ROWS
    lds = np.zeros(RING * SLAB_BYTES // 2, np.uint16)
    acc = np.zeros((8, 4, 2, 64, 16), np.int64)                   # wave, a, b, lane, reg
    for q in range(P):
        slot = q % RING
        # ---- staging (Stager::issue): wave w, lane l
        for w in range(8):
            for l in range(64):
                st_row = 8 * (w >> 2) + ((l >> 2) & 7)
                st_col = 64 * (w & 3) + 32 * (l >> 5) + 8 * (l & 3)
                row = q * SLAB_ROWS + st_row
                dst = slot * SLAB_BYTES + w * 1024 + l * 16        # lane-linear
                va = dy16[row, n0 + st_col:n0 + st_col + 8] if row < M else np.zeros(8, np.uint16)
                vb = x16[row, k0 + st_col:k0 + st_col + 8] if row < M else np.zeros(8, np.uint16)
                lds[dst // 2:dst // 2 + 8] = va
                lds[(dst + 8192) // 2:(dst + 8192) // 2 + 8] = vb
        # ---- reads + MFMA
        for w in range(8):
            wr, wc = w >> 2, w & 3
            lane = np.arange(64)
            gq, p16 = lane >> 4, lane & 15
            lane_off = (4 * (gq >> 1) + (p16 >> 2)) * 64 + (gq & 1) * 32 + (p16 & 3) * 8
            a_off = lane_off + 4 * wr * 512
            b_off = lane_off + 8192 + 2 * wc * 512
            base = slot * SLAB_BYTES

            def frag(off, i):
                lo = tr_read(lds, base + off + i * 512)
                hi = tr_read(lds, base + off + 4096 + i * 512)
                return np.concatenate([lo, hi], axis=1).view(np.int16).astype(np.int64)   # [64, 8]
            fa = [frag(a_off, i) for i in range(4)]
            fb = [frag(b_off, i) for i in range(2)]
            # v_mfma_f32_32x32x16: A lane l = row l&31, k = 8(l>>5)+e; B lane l = col l&31, same k;
            # D lane l reg r: col l&31, row (r&3) + 8(r>>2) + 4(l>>5)
            for a in range(4):
                for b in range(2):
                    A = np.zeros((32, 16), np.int64); B = np.zeros((16, 32), np.int64)
                    for l in range(64):
                        for e in range(8):
                            A[l & 31, 8 * (l >> 5) + e] = fa[a][l, e]
                            B[8 * (l >> 5) + e, l & 31] = fb[b][l, e]
                    D = A @ B
                    for l in range(64):
                        for r in range(16):
                            acc[w, a, b, l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    # ---- epilogue (flush_tile): through a [32][32] scratch, then rows 8j + lane>>3, columns 4(lane&7)..+3
    dw = np.zeros((512, 512), np.int64)
    for w in range(8):
        wr, wc = w >> 2, w & 3
        n_base, k_base = n0 + 128 * wr, k0 + 64 * wc
        for a in range(4):
            for b in range(2):
                sw = np.zeros(1024, np.int64)
                for l in range(64):
                    for r in range(16):
                        sw[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[w, a, b, l, r]
                for l in range(64):
                    rr, cq = l >> 3, l & 7
                    for j in range(4):
                        t = sw[(8 * j + rr) * 32 + 4 * cq:(8 * j + rr) * 32 + 4 * cq + 4]
                        dw[n_base + 32 * a + 8 * j + rr, k_base + 32 * b + 4 * cq:k_base + 32 * b + 4 * cq + 4] += t
    ref = dy[:, n0:n0 + 256].T.astype(np.int64) @ x[:, k0:k0 + 256].astype(np.int64)
    got = dw[n0:n0 + 256, k0:k0 + 256]
    ok = np.array_equal(ref, got)
    print('wgrad256 emulation M=%d: %s (max |diff| %d)' % (M, 'OK' if ok else 'MISMATCH', np.abs(ref - got).max()))
    assert (dw.sum() == got.sum())
    return ok


if __name__ == '__main__':
    assert main(40) and main(16, seed=1) and main(200, n0=0, k0=0, seed=2)

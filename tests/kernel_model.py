"""Lane-level numpy model of the data movement in aac_synth.hip and fir.hip.

Not a compute path: it re-enacts, lane by lane, exactly the index arithmetic the HIP
kernels use (register slots, LDS exchange addresses, MFMA fragment layouts) so that the
mapping can be checked against the oracle on a machine without a GPU.
"""
import numpy as np

LANES = np.arange(64)
HI3, LO3 = LANES >> 3, LANES & 7


def twiddle(n):
    """dsp.rs:99-106 in f64 (the model checks indexing, not rounding)."""
    b = np.arange(n // 2)
    ang = np.pi / n * (b + 0.125)
    return (np.cos(ang) + 1j * np.sin(ang)) * np.sqrt((1.0 / 32768.0) / n)


def dft8(z):
    """z[8][64]: forward DFT along the register axis."""
    return np.fft.fft(z, axis=0)


def long_imdct_positions(X):
    """Returns (im1[16][64], im2[16][64]): imdct[i] and imdct[1024+i] at each lane's slots."""
    X = np.asarray(X, np.float64)
    tw = twiddle(1024)
    xin_x = np.stack([X[128 * r + 2 * LANES] for r in range(8)])      # [r][lane]
    xin_y = np.stack([X[128 * r + 2 * LANES + 1] for r in range(8)])
    z = np.zeros((8, 64), complex)
    for r in range(8):
        even = xin_x[r]
        odd = -xin_y[7 - r][63 - LANES]  # __shfl(x[7-r].y, 63 - lane)
        t = tw[LANES + 64 * r]
        z[r] = (odd * t.imag - even * t.real) + 1j * (odd * t.real + even * t.imag)
    z = dft8(z)
    w64 = np.exp(-2j * np.pi * np.arange(64) / 64)
    w512 = np.exp(-2j * np.pi * np.arange(512) / 512)
    ex = np.zeros(576, complex)
    for k in range(8):
        z[k] = z[k] * w64[(HI3 * k) & 63]
    for k in range(8):
        ex[k * 68 + LANES] = z[k]
    for n2 in range(8):
        z[n2] = ex[LO3 * 68 + 8 * n2 + HI3]
    z = dft8(z)
    for k in range(8):
        z[k] = z[k] * w512[(HI3 * (LO3 + 8 * k)) & 511]
    ex[:] = 0
    for k in range(8):
        ex[HI3 * 72 + k * 8 + LO3] = z[k]
    for n3 in range(8):
        z[n3] = ex[n3 * 72 + LANES]
    z = dft8(z)  # z[j][lane] = Z[lane + 64 j]
    ex[:] = 0
    for j in range(8):
        ex[LANES + 64 * j] = tw[LANES + 64 * j] * np.conj(z[j])
    im1 = np.zeros((16, 64))
    im2 = np.zeros((16, 64))
    for r in range(2):
        q = 2 * LANES + 128 * r
        F0, F1, M0, M1 = ex[256 + q], ex[257 + q], ex[254 - q], ex[255 - q]
        Fx, Fy, Fz, Fw = F0.real, F0.imag, F1.real, F1.imag
        Mx, My, Mz, Mw = M0.real, M0.imag, M1.real, M1.imag
        im1[8 * r + 0], im1[8 * r + 1], im1[8 * r + 2], im1[8 * r + 3] = -Fx, -Mw, -Fz, -My
        im1[8 * r + 4], im1[8 * r + 5], im1[8 * r + 6], im1[8 * r + 7] = My, Fz, Mw, Fx
        im2[8 * r + 0], im2[8 * r + 1], im2[8 * r + 2], im2[8 * r + 3] = Fy, Mz, Fw, Mx
        im2[8 * r + 4], im2[8 * r + 5], im2[8 * r + 6], im2[8 * r + 7] = Mx, Fw, Mz, Fy
    return im1, im2


def slot_positions():
    """pos[s][lane]: sample index held in slot s of each lane."""
    pos = np.zeros((16, 64), int)
    for r in range(2):
        for c in range(4):
            pos[8 * r + c] = 4 * LANES + 256 * r + c
            pos[8 * r + 4 + c] = 1020 - 4 * LANES - 256 * r + c
    return pos


def long_imdct(X):
    """2048-sample IMDCT assembled from the per-lane slots."""
    im1, im2 = long_imdct_positions(X)
    pos = slot_positions()
    out = np.zeros(2048)
    out[pos] = im1
    out[1024 + pos] = im2
    return out


def short_spectra(X):
    """ex[64 w + k] = post-twiddled spectrum of short block w, as the eight-short path leaves it in LDS."""
    X = np.asarray(X, np.float64)
    tws = twiddle(128)
    w64 = np.exp(-2j * np.pi * np.arange(64) / 64)
    ex = np.zeros(576, complex)
    for w in range(8):
        x_x = X[128 * w + 2 * LANES]
        x_y = X[128 * w + 2 * LANES + 1]
        even = x_x
        odd = -x_y[63 - LANES]
        t = tws[LANES]
        ex[64 * w + LANES] = (odd * t.imag - even * t.real) + 1j * (odd * t.real + even * t.imag)
    g = np.zeros((8, 64), complex)
    for b in range(8):
        g[b] = ex[64 * HI3 + LO3 + 8 * b]
    g = dft8(g)
    for kb in range(1, 8):
        g[kb] = g[kb] * w64[LO3 * kb]
    for kb in range(8):
        ex[64 * HI3 + 8 * kb + LO3] = g[kb]
    for aa in range(8):
        g[aa] = ex[64 * HI3 + 8 * LO3 + aa]
    g = dft8(g)
    for ka in range(8):
        k = LO3 + 8 * ka
        ex[64 * HI3 + k] = tws[k] * np.conj(g[ka])
    return ex


def short_sample(v, t):
    seg, u = t >> 6, t & 63
    odd = u & 1
    lo = (63 - u) >> 1 if odd else u >> 1
    if seg == 0:
        return -v[lo].imag if odd else -v[32 + lo].real
    if seg == 1:
        return v[32 + lo].real if odd else v[lo].imag
    if seg == 2:
        return v[lo].real if odd else v[32 + lo].imag
    return v[32 + lo].imag if odd else v[lo].real


def short_buffer(X, prev_short, cur_short):
    ex = short_spectra(X)
    buf = np.zeros(2048)
    for p in range(2048):
        q = p - 448
        if q < 0 or q >= 1152:
            continue
        hi = q >> 7
        acc = 0.0
        for d in (1, 0):
            w = hi - d
            if w < 0 or w > 7:
                continue
            t = q - 128 * w
            win = prev_short[t] if (w == 0 and t < 128) else cur_short[t]
            acc += short_sample(ex[64 * w:64 * w + 64], t) * win
        buf[p] = acc
    return buf


# ---- FIR (fir.hip) ----------------------------------------------------------------------------

def fir_afrag(taps):
    af = np.zeros((76, 64))
    for s in range(76):
        for l in range(64):
            p = 16 * (s >> 2) + 4 * (l >> 4) + (s & 3) - 3 * (l & 15) - 3
            if 0 <= p < 256:
                af[s, l] = taps[p]
    return af


def mfma_16x16x4(a_frag, b_frag, acc):
    """v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[i=4*(l>>4)+r][j=l&15] in acc[r][l]."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a_frag[l]
        B[l >> 4, l & 15] = b_frag[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for r in range(4):
            out[r, l] += D[4 * (l >> 4) + r, l & 15]
    return out


def fir_model(x_rows, taps, out_first, out_count, in_origin=0):
    """Runs the kernel's schedule for one 16-row group and one segment covering all blocks."""
    x_rows = np.asarray(x_rows, np.float64)
    rows, in_frames = x_rows.shape
    assert rows <= 16
    af = fir_afrag(taps)
    t0 = 3 * out_first - 128
    total_blocks = (out_count + 15) // 16
    out = np.zeros((rows, out_count))

    def sample(row, n2):  # local sample n'' of a row, zero outside the input
        idx = n2 + t0 - in_origin
        if row >= rows or idx < 0 or idx >= in_frames:
            return 0.0
        return x_rows[row, idx]

    acc = [np.zeros((4, 64)) for _ in range(7)]
    lanes = np.arange(64)
    j, kq = lanes & 15, lanes >> 4
    for A in range(0, total_blocks + 6):
        for gi in range(3):
            G = 3 * A + gi
            xb = np.zeros((4, 64))
            for t in range(4):
                for l in range(64):
                    xb[t, l] = sample(j[l], 16 * G + 4 * kq[l] + t)
            for t in range(4):
                u = 4 * gi + t
                for b in range(7):
                    if 12 * b + u < 76:
                        acc[b] = mfma_16x16x4(af[12 * b + u], xb[t], acc[b])
            if gi == 0:
                blk = A - 6
                if 0 <= blk < total_blocks:
                    for l in range(64):
                        if j[l] < rows:
                            for r in range(4):
                                m = blk * 16 + 4 * kq[l] + r
                                if m < out_count:
                                    out[j[l], m] = acc[6][r, l]
        acc = [np.zeros((4, 64))] + acc[:6]
    return out

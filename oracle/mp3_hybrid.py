"""oracle/mp3_hybrid.py -- CPU restatement (numpy, f64) of the MPEG-1/2 Layer III hybrid synthesis filterbank.

TEST INFRASTRUCTURE ONLY; nothing under soundkit_amd/ imports it.

PARITY UNPINNED.  In the reference this arithmetic lives in a third-party crate, `nanomp3` (soundkit-mp3/Cargo.toml:13:
git https://github.com/wavey-ai/nanomp3.git, branch main, no pinned revision), reached through
`nanomp3::Decoder::decode` (soundkit-mp3/src/lib.rs:284, 312, 340).  Its source is not in the reference tree, there is
no decoded-PCM golden for MP3, and two data tables of ISO/IEC 11172-3 that a whole decoder needs -- the
512-coefficient synthesis window D[] (Table B.3) and the 32 Huffman tables (B.7) -- exist nowhere in this container and
have no closed form.  What IS closed-form is restated here from the published algorithm, ISO/IEC 11172-3:1993
2.4.3.4 (and the reference decoder's arrangement of it):

  alias reduction      2.4.3.4.10.1 / Table B.9: 8 butterflies per subband boundary, cs_i = 1/sqrt(1+c_i^2),
                       ca_i = c_i/sqrt(1+c_i^2), c = -0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037
  IMDCT                2.4.3.4.10.2: n = 36 (block types 0, 1, 3) or three n = 12 transforms (block type 2):
                       x_i = sum_k X_k cos(pi/(2n) (2i + 1 + n/2)(2k + 1))
  windows              2.4.3.4.10.3: the four block-type windows (sine halves of length 36 and 12, with the flat / zero runs)
  overlap-add          2.4.3.4.10.4: first half + stored second half of the previous granule
  frequency inversion  2.4.3.4.10.5 (compensation for the polyphase filterbank): odd samples of odd subbands negated
  polyphase synthesis  2.4.3.2 / figure A.2: V shifted by 64, V_i = sum_k cos((16+i)(2k+1) pi/64) S_k, U from V,
                       W = U * D, 32 outputs = sums of 16

The synthesis window D is an INPUT (sk_mp3_set_synthesis_window on the product side): with the real Table B.3 these
functions are the standard's filterbank; the tests use a synthetic prototype, which checks the arithmetic, not the table.
Input convention for block type 2: the 18 values of a subband are interleaved by window, X_w[m] = xr[18 sb + 3 m + w]
(the order the standard's reordering step leaves them in).
"""
import numpy as np

C_ALIAS = np.array([-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037])  # ISO 11172-3 Table B.9
CS = 1.0 / np.sqrt(1.0 + C_ALIAS ** 2)
CA = C_ALIAS / np.sqrt(1.0 + C_ALIAS ** 2)


def block_window(block_type):
    """2.4.3.4.10.3 -> 36 values (block types 0, 1, 3) or 12 (block type 2)"""
    i = np.arange(36)
    if block_type == 0:
        return np.sin(np.pi / 36 * (i + 0.5))
    if block_type == 1:
        w = np.zeros(36)
        w[:18] = np.sin(np.pi / 36 * (i[:18] + 0.5))
        w[18:24] = 1.0
        w[24:30] = np.sin(np.pi / 12 * (i[24:30] - 18 + 0.5))
        return w
    if block_type == 3:
        w = np.zeros(36)
        w[6:12] = np.sin(np.pi / 12 * (i[6:12] - 6 + 0.5))
        w[12:18] = 1.0
        w[18:] = np.sin(np.pi / 36 * (i[18:] + 0.5))
        return w
    return np.sin(np.pi / 12 * (np.arange(12) + 0.5))


def imdct(x, n):
    """2.4.3.4.10.2: n/2 inputs -> n outputs"""
    i = np.arange(n)[:, None]
    k = np.arange(n // 2)[None, :]
    return (np.cos(np.pi / (2 * n) * (2 * i + 1 + n // 2) * (2 * k + 1)) * np.asarray(x, np.float64)[None, :]).sum(axis=1)


def subband_block(x18, block_type):
    """one subband of one granule: 18 frequency lines -> 36 windowed time samples"""
    if block_type != 2:
        return imdct(x18, 36) * block_window(block_type)
    out = np.zeros(36)
    win = block_window(2)
    for w in range(3):
        out[6 * w + 6:6 * w + 18] += imdct(np.asarray(x18, np.float64)[w::3], 12) * win
    return out


def alias_reduce(xr, block_type, mixed):
    """2.4.3.4.10.1: every boundary for long blocks, the first one only for mixed blocks, none for short blocks"""
    xr = np.array(xr, np.float64)
    if block_type == 2 and not mixed:
        return xr
    for sb in range(1, 2 if block_type == 2 else 32):
        for i in range(8):
            lo, hi = xr[18 * sb - 1 - i], xr[18 * sb + i]
            xr[18 * sb - 1 - i] = lo * CS[i] - hi * CA[i]
            xr[18 * sb + i] = hi * CS[i] + lo * CA[i]
    return xr


MATRIX = np.cos((16 + np.arange(64))[:, None] * (2 * np.arange(32)[None, :] + 1) * np.pi / 64)  # 2.4.3.2 N_ik


class Channel:
    """the carried state of one channel: the IMDCT overlap (32 x 18) and the polyphase FIFO V (1024)"""

    def __init__(self):
        self.overlap = np.zeros((32, 18))
        self.v = np.zeros(1024)

    def hybrid(self, xr, block_type, mixed):
        """576 frequency lines -> hybrid[sb][ss], the 18 time samples of each of the 32 subbands"""
        xr = alias_reduce(xr, block_type, mixed)
        out = np.zeros((32, 18))
        for sb in range(32):
            bt = 0 if (block_type == 2 and mixed and sb < 2) else block_type
            raw = subband_block(xr[18 * sb:18 * sb + 18], bt)
            out[sb] = raw[:18] + self.overlap[sb]
            self.overlap[sb] = raw[18:]
        out[1::2, 1::2] *= -1.0   # frequency inversion
        return out

    def polyphase(self, s32, d512):
        """one time slot: 32 subband samples -> 32 PCM samples (figure A.2)"""
        self.v[64:] = self.v[:-64].copy()
        self.v[:64] = MATRIX @ np.asarray(s32, np.float64)
        u = np.zeros(512)
        for i in range(8):
            u[64 * i:64 * i + 32] = self.v[128 * i:128 * i + 32]
            u[64 * i + 32:64 * i + 64] = self.v[128 * i + 96:128 * i + 128]
        w = u * d512
        return w.reshape(16, 32).sum(axis=0)

    def granule(self, xr, block_type, mixed, d512):
        """576 frequency lines -> 576 PCM samples"""
        h = self.hybrid(xr, block_type, mixed)
        return np.concatenate([self.polyphase(h[:, ss], d512) for ss in range(18)])


def synthetic_window(seed=0):
    """A stand-in for Table B.3 with its structure (a long low-pass prototype whose 64-sample segments alternate in
    sign, as cosine modulation needs) but NOT its values: the tests check the arithmetic around D, not D."""
    n = np.arange(512)
    proto = np.sinc((n - 255.5) / 64.0) * np.hanning(512) / 32.0
    signs = np.where((n // 64) % 2 == 1, -1.0, 1.0)
    jitter = 1.0 + 0.01 * np.random.default_rng(seed).standard_normal(512)
    return proto * signs * jitter

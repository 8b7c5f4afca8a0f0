// mp3_requant.hip -- Layer III requantisation, joint-stereo processing and the short-block reorder for gfx950, batched
// over granules: what sits between the Huffman stage of nanomp3::Decoder::decode (soundkit-mp3/src/lib.rs:284; the
// crate's source is not in the reference tree) and the hybrid synthesis of mp3_hybrid.hip.  ISO/IEC 11172-3 in closed form:
//   2.4.3.4.7.1  xr = sign(is) |is|^(4/3) 2^(q / 4),
//                q = global_gain - 210 - 8 subblock_gain[w] - (scalefac_scale ? 4 : 2) (scalefac + preflag pretab)
//   2.4.3.4.9    mid/side: L = (M + S) / sqrt 2, R = (M - S) / sqrt 2
//   2.4.3.4.9.3  MPEG-1 intensity: in the bands above the last one where the right channel holds anything (per window for
//                short blocks) the right channel's scale factor is a position is_pos; L = xr k, R = xr (1 - k),
//                k = t / (1 + t), t = tan(is_pos pi / 12); is_pos 7 = "not intensity coded".  Which bands those are in a mixed
//                granule, and what the last band (it has no factor of its own) takes, follows minimp3 -- see the kernel
//   13818-3 2.4.3.2  MPEG-2 / 2.5 intensity: the same bands, ratios 1 : i0^n or i0^n : 1 (i0 = 2^-1/4 or 2^-1/2 by intensity_scale);
//                the position that means "not intensity coded" is the largest value its field holds -- the host marks it (bit 7)
//   2.4.3.4.8    short blocks: band-by-band [window][line] -> [line][window]
// The scale-factor band offsets (Table B.8) and the pre-emphasis table are set per engine and rate (sk_mp3_set_band_tables;
// sk_mp3_decoder_create installs the standard's, csrc/mp3_iso_tables.h).  oracle/mp3_bitstream.py (requantize_granule) is the f64 checker; what pins the MP3 row: DESIGN.md section 2.
//
// One wavefront per granule, four per block.  A lane takes the lines lane + 64 k (k < 9) of both channels in bitstream
// order: band, window and destination from the engine's line map, requantised into registers; the highest
// occupied band of the right channel is a wave maximum; the stereo step works on the lane's own pair of values; the integers
// come in and the reordered lines go out through LDS, so that both cross HBM as whole 256-byte runs.  Traffic per granule-channel:
// 1152 B in, 2304 B out; the kernel is HBM-bound like everything else on this path.
#include "sk_device.h"

namespace sk {

namespace {

constexpr int kWaves = 4;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

struct Line {  // where a bitstream-order line sits (unpacked from the engine's line map: dest | band << 10 | (win + 1) << 15)
    int band;  // long band 0..21 or short band 0..12
    int win;   // -1: long
    int dest;  // its position after the reorder
};
__device__ __forceinline__ Line unpack_line(uint32_t m) {
    Line r;
    r.dest = (int)(m & 1023u);
    r.band = (int)((m >> 10) & 31u);
    r.win = (int)((m >> 15) & 3u) - 1;
    return r;
}

}  // namespace

// One wavefront per granule, four per block.  Where a line sits -- band, window, position after the reorder -- depends on the
// sampling rate and on how the granule is cut up (long, short, mixed) only: the engine tabulates it once per rate
// (Mp3RequantArgs::line_map, 3 x 576 words, read coalesced), and the exponent q of every band / window of the granule is
// worked out once by 61 lanes into LDS; a line then costs one map word, one LDS read for q, one gather of |is|^(4/3).
__global__ __launch_bounds__(kWaves * 64) void k_mp3_requant(Mp3RequantArgs a) {
    // the granule's integers (brought in as whole dwords) and, once they have all been read, its reordered lines share one buffer:
    // 5 KB of LDS per wave instead of 7, 32 waves per CU instead of 20 -- the kernel lives on how many of its chains of
    // dependent memory trips are in flight
    __shared__ __attribute__((aligned(16))) float xs[kWaves][2][576];
    __shared__ uint32_t rec[kWaves][sizeof(Mp3RequantRecord) / 4];
    __shared__ int16_t qtab[kWaves][2][64];  // [channel][long band 0..21 | 22 + 3 band + window]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gi = blockIdx.x * kWaves + wave;
    if (gi >= a.n) return;
    {
        const uint32_t *src = (const uint32_t *)(a.records + gi);
        if (lane < (int)(sizeof(Mp3RequantRecord) / 4)) rec[wave][lane] = src[lane];
    }
    wave_sync();
    const Mp3RequantRecord &g = *(const Mp3RequantRecord *)rec[wave];
    const int channels = g.channels;
    float *out = a.xr + (size_t)g.off * 576;
    if (g.slot >= kMp3Rates) {  // rejected on the host: silence, so that the synthesis after it stays defined
        for (int c = 0; c < channels; ++c)
            for (int k = 0; k < 9; ++k) out[c * 576 + lane + 64 * k] = 0.0f;
        return;
    }
    {
        // 1152 bytes per channel in 256-byte runs (a lane's own lines are 64 apart: read one by one they are 2-byte gathers)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.is + (size_t)g.off * 576);  // 1152-byte granularity: dword aligned
        uint32_t *dst = reinterpret_cast<uint32_t *>(xs[wave]);
        for (int d = lane; d < channels * 288; d += 64) dst[d] = src[d];
    }
    for (int c = 0; c < channels; ++c) {  // q of every band (long) and band x window (short): 2.4.3.4.7.1
        const sk_mp3_requant_channel &ch = g.ch[c];
        const int mult = ch.scalefac_scale ? 4 : 2;
        int q = (int)ch.global_gain - 210;
        // (bit 7 of a scale factor marks an LSF intensity position as "not intensity coded", below; the factor is the low bits)
        if (lane < 22) q -= mult * ((int)(ch.scalefac_l[lane] & 0x7f) + (ch.preflag ? (int)a.pretab[g.slot * 24 + lane] : 0));
        else if (lane < 61) q -= 8 * (int)ch.subblock_gain[(lane - 22) % 3] + mult * (int)(ch.scalefac_s[(lane - 22) / 3][(lane - 22) % 3] & 0x7f);
        qtab[wave][c][lane] = (int16_t)q;
    }
    wave_sync();
    const int16_t *is = reinterpret_cast<const int16_t *>(xs[wave]);

    // Stereo tools as minimp3 does them (nanomp3, the reference's decoder, is its port; L3_stereo_top_band / L3_intensity_stereo /
    // L3_stereo_process): the bands of a granule are numbered in the order their scale factors come -- long bands, then short
    // bands band by band, window by window -- r = band | 3 band + window for long and short granules; a mixed granule: its long
    // bands, then 64 + 3 band + window (only the order matters there, whatever band tables the engine was given).
    // bound[r % 3] = the highest r holding a non-zero line of the right channel; long AND mixed granules use the largest of the
    // three for every band, short granules one per window.  Band r is intensity coded iff r > bound and its position is a legal one.
    const int layout0 = g.ch[0].block_type == 2 ? (g.ch[0].mixed_block_flag ? 2 : 1) : 0;  // both channels agree whenever this is used (host check)
    auto running = [&](const Line &at) { return at.win < 0 ? at.band : (layout0 == 2 ? 64 : 0) + 3 * at.band + at.win; };
    float v[2][9];
    Line where[2][9];
    int bound[3] = {-1, -1, -1};
    for (int c = 0; c < channels; ++c) {
        const sk_mp3_requant_channel &ch = g.ch[c];
        const int layout = ch.block_type == 2 ? (ch.mixed_block_flag ? 2 : 1) : 0;
        const uint32_t *map = a.line_map + ((size_t)g.slot * 3 + layout) * 576;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = lane + 64 * k;
            const Line at = unpack_line(map[i]);
            where[c][k] = at;
            const int q_in = is[c * 576 + i];
            const int q = qtab[wave][c][at.win < 0 ? at.band : 22 + 3 * at.band + at.win];
            const int mag = min(abs(q_in), (int)kMp3Pow43 - 1);
            const float m = a.pow43[mag] * a.root4[q & 3];  // 2^(q / 4) = 2^floor(q / 4) * 2^((q mod 4) / 4): one rounding here,
            const float x = ldexpf(m, q >> 2);              // the power of two is exact
            v[c][k] = q_in < 0 ? -x : x;
            if (c == 1 && q_in != 0) {
                const int r = running(at), m = r % 3;
                bound[0] = m == 0 ? max(bound[0], r) : bound[0];
                bound[1] = m == 1 ? max(bound[1], r) : bound[1];
                bound[2] = m == 2 ? max(bound[2], r) : bound[2];
            }
        }
    }

    if (channels == 2 && (g.flags & 3)) {
        const bool ms = g.flags & 1, intensity = g.flags & 2, lsf = g.flags & 4;
        const int lsf_shift = (g.flags >> 3) & 1;  // 13818-3: intensity_scale, the low bit of the right channel's scalefac_compress
        if (intensity) {
            for (int w = 0; w < 3; ++w) bound[w] = wave_max(bound[w]);
            if (layout0 != 1) bound[0] = bound[1] = bound[2] = max(max(bound[0], bound[1]), bound[2]);
        }
        const sk_mp3_requant_channel &right = g.ch[1];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const Line at = where[0][k];
            bool done = false;
            if (intensity) {
                const int r = running(at);
                const int limit = r % 3 == 0 ? bound[0] : (r % 3 == 1 ? bound[1] : bound[2]);
                if (r > limit) {
                    // The last band (long 21, short 12) has no scale factor of its own: it takes the position of the band below in its
                    // window -- if that one is intensity coded itself; otherwise (the bound reaches it: its factor is a scale factor,
                    // not a position) the position that leaves both channels alike: 3 (MPEG-1), 0 (13818-3).
                    const bool last = at.win < 0 ? at.band >= 21 : at.band >= 12;
                    const int below = r - (at.win < 0 ? 1 : 3);
                    int pos;
                    if (last && limit >= below) pos = lsf ? 0 : 3;
                    else pos = at.win < 0 ? right.scalefac_l[min(at.band, 20)] : right.scalefac_s[min(at.band, 11)][at.win];
                    if (lsf) {
                        // ISO/IEC 13818-3 2.4.3.2: is_pos = 0: both channels get the line; odd: the left one is scaled by
                        // i0^((is_pos + 1) / 2), even: the right one by i0^(is_pos / 2), i0 = 2^-1/4 or 2^-1/2 (intensity_scale).
                        // A position equal to the largest value its field can hold means "not intensity coded" (host: bit 7).
                        if (!(pos & 0x80)) {
                            const int steps = ((pos + 1) >> 1) << lsf_shift;  // in units of 2^-1/4
                            const float f = ldexpf(a.root4[(4 - (steps & 3)) & 3], -((steps + 3) >> 2));  // 2^(-steps / 4): one table rounding
                            const float x = v[0][k];
                            v[0][k] = (pos & 1) ? x * f : x;
                            v[1][k] = (pos & 1) ? x : x * f;
                            done = true;
                        }
                    } else if (pos < 7) {
                        const float kl = a.is_k[pos];
                        const float x = v[0][k];
                        v[0][k] = x * kl;
                        v[1][k] = x * (1.0f - kl);
                        done = true;
                    }
                }
            }
            if (!done && ms) {
                const float m = v[0][k], s = v[1][k];
                v[0][k] = (m + s) * 0.70710678118654752440f;
                v[1][k] = (m - s) * 0.70710678118654752440f;
            }
        }
    }

    wave_sync();  // every integer has been read: the buffer changes hands
    for (int c = 0; c < channels; ++c) {
#pragma unroll
        for (int k = 0; k < 9; ++k) xs[wave][c][where[c][k].dest] = v[c][k];
    }
    wave_sync();
    for (int c = 0; c < channels; ++c) {
#pragma unroll
        for (int k = 0; k < 9; ++k) out[c * 576 + lane + 64 * k] = xs[wave][c][lane + 64 * k];
    }
}

hipError_t launch_mp3_requant(const Mp3RequantArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    k_mp3_requant<<<(a.n + kWaves - 1) / kWaves, kWaves * 64, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace sk

// mp3_hybrid.hip -- the MPEG-1/2 Layer III hybrid synthesis filterbank for gfx950, batched over streams.
//
// In the reference this is the tail of nanomp3::Decoder::decode (soundkit-mp3/src/lib.rs:284; the crate's source is not
// in the tree), followed by f32_to_i16 (lib.rs:376-385).  What is built here is what ISO/IEC 11172-3 2.4.3.4 defines in
// closed form: alias reduction, IMDCT 36 / 3 x 12 with the four block-type windows, overlap-add, frequency inversion and
// the 32-band polyphase synthesis.  The 512-coefficient synthesis window D (Table B.3) is data this container does not
// hold: the engine takes it from the caller (sk_mp3_set_synthesis_window).  oracle/mp3_hybrid.py is the f64 checker.
//
// One wavefront owns one (stream, channel) and walks its granules; the carried state (IMDCT overlap 32 x 18, the
// polyphase FIFO V as a ring of 16 x 64, the ring position) crosses HBM once per launch.  Per granule:
//   x[576] -> LDS; 8 alias butterflies per subband boundary in place (lanes = butterflies);
//   IMDCT + window as ONE 36 x 18 matrix per block type (block type 2's three short transforms are a sparse matrix of
//   the same shape): lane = subband + 32 * half computes 18 of the 36 outputs, matrix rows read as LDS broadcasts;
//   first half + stored overlap -> frequency inversion -> hyb[ss][sb] in LDS, second half -> the new overlap;
//   18 time slots: V_i = sum_k N[i][k] S_k with lane i holding row i of N in 32 VGPRs and S_k read as broadcasts,
//   into the ring; every lane then sums its 8 window taps  ring[(p - 2 i' - half) & 15][lane] * D[64 i' + lane]  and
//   lanes j, j + 32 add up to output j.
#include "sk_device.h"

namespace sk {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f4 lds_f4;

constexpr int kWaves = 4;
constexpr int kRow = 20;  // matrix rows padded from 18 to 20 floats: 16-byte aligned ds_read_b128

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// soundkit-mp3/src/lib.rs:376-385: (x * 32767).round(), saturating
__device__ __forceinline__ int16_t mp3_to_i16(float x) {
    const float scaled = roundf(x * 32767.0f);
    if (scaled != scaled) return 0;  // NaN: Rust's saturating `as` gives 0
    if (scaled > 32767.0f) return 32767;
    if (scaled < -32768.0f) return -32768;
    return (int16_t)scaled;
}

template <bool OUT16>
__global__ __launch_bounds__(kWaves * 64, 3) void k_mp3_hybrid(Mp3Args a) {
    __shared__ __attribute__((aligned(16))) float mat[4 * 36 * kRow];
    // the granule's lines and, once every lane holds its eighteen of them, the hybrid samples share one buffer: 8.7 KB of LDS per
    // wave instead of 11, three workgroups per CU instead of two (the kernel is a chain of LDS round trips: it lives on waves)
    __shared__ __attribute__((aligned(16))) float xs[kWaves][576];
    __shared__ __attribute__((aligned(16))) float ovl[kWaves][576];
    __shared__ __attribute__((aligned(16))) float ring[kWaves][1024];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4 * 36 * kRow; i += kWaves * 64) mat[i] = a.imdct[i];
    __syncthreads();
    const uint32_t task_id = blockIdx.x * kWaves + wave;
    if (task_id >= a.n_tasks) return;
    const SynthTask task = a.tasks[task_id];
    const uint32_t count = __builtin_amdgcn_readfirstlane(task.count), state = __builtin_amdgcn_readfirstlane(task.state);
    const SynthEntry *entries = a.entries + __builtin_amdgcn_readfirstlane(task.begin);
    lds_f *x = (lds_f *)xs[wave], *h = (lds_f *)xs[wave], *ov = (lds_f *)ovl[wave], *rg = (lds_f *)ring[wave];
    const lds_f *m = (const lds_f *)mat;

    // carried state in: overlap[576], ring[1024], position
    float *st = a.state + (size_t)state * kMp3StateFloats;
    for (int i = lane; i < 576; i += 64) ov[i] = st[i];
    for (int i = lane; i < 1024; i += 64) rg[i] = st[576 + i];
    uint32_t pos = __builtin_amdgcn_readfirstlane(__float_as_uint(st[1600])) & 15u;
    // per-lane constants: row `lane` of the matrixing N, and this lane's 8 window taps
    float nrow[32], dwin[8];
#pragma unroll
    for (int k = 0; k < 32; ++k) nrow[k] = a.matrix[lane * 32 + k];
#pragma unroll
    for (int i = 0; i < 8; ++i) dwin[i] = a.window[64 * i + lane];
    const int sb = lane & 31, half = lane >> 5;
    wave_sync();

    for (uint32_t g = 0; g < count; ++g) {
        const SynthEntry ent = entries[g];
        const uint32_t win = __builtin_amdgcn_readfirstlane(ent.win);
        const int block_type = win & 3, mixed = (win >> 2) & 1, channels = ((win >> 3) & 1) + 1, ch = (win >> 4) & 1;
        const size_t off = (size_t)__builtin_amdgcn_readfirstlane(ent.off1024);  // in units of 576 floats
        const float *src = a.xr + off * 576;
        for (int i = lane; i < 576; i += 64) x[i] = src[i];
        wave_sync();
        // ---- alias reduction (2.4.3.4.10.1) ----
        const int boundaries = block_type == 2 ? (mixed ? 1 : 0) : 31;
        for (int b = lane; b < boundaries * 8; b += 64) {
            const int s = (b >> 3) + 1, i = b & 7;
            const float lo = x[18 * s - 1 - i], hi = x[18 * s + i];
            const float cs = a.cs_ca[i], ca = a.cs_ca[8 + i];
            x[18 * s - 1 - i] = lo * cs - hi * ca;
            x[18 * s + i] = hi * cs + lo * ca;
        }
        wave_sync();
        // ---- IMDCT + window: out[i] = sum_k M[bt][18 half + i][k] x[18 sb + k] ----
        float in[18], out[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) in[k] = x[18 * sb + k];
        const int bt = (block_type == 2 && mixed && sb < 2) ? 0 : block_type;
        const lds_f *rows = m + (bt * 36 + 18 * half) * kRow;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 5; ++k4) {
                const f4 c = *reinterpret_cast<const lds_f4 *>(rows + i * kRow + 4 * k4);
                acc += c.x * in[4 * k4];
                acc += c.y * in[4 * k4 + 1];
                if (k4 < 4) {
                    acc += c.z * in[4 * k4 + 2];
                    acc += c.w * in[4 * k4 + 3];
                }
            }
            out[i] = acc;
        }
        // ---- overlap-add, frequency inversion; the second half becomes the overlap ----
        wave_sync();  // every lane has read its lines: the buffer changes hands
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                float v = out[i] + ov[18 * sb + i];
                if ((sb & 1) && (i & 1)) v = -v;
                h[32 * i + sb] = v;
            }
        }
        wave_sync();
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < 18; ++i) ov[18 * sb + i] = out[i];
        }
        wave_sync();
        // ---- polyphase synthesis, 18 time slots ----
        for (int ss = 0; ss < 18; ++ss) {
            pos = (pos + 1u) & 15u;  // the new vector takes the place of the oldest
            float v = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f4 s4 = *reinterpret_cast<const lds_f4 *>(h + 32 * ss + 4 * k4);
                v += nrow[4 * k4] * s4.x;
                v += nrow[4 * k4 + 1] * s4.y;
                v += nrow[4 * k4 + 2] * s4.z;
                v += nrow[4 * k4 + 3] * s4.w;
            }
            rg[64 * pos + lane] = v;
            wave_sync();
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += rg[64 * ((pos - 2u * (uint32_t)i - (uint32_t)half) & 15u) + lane] * dwin[i];
            const float sum = acc + __shfl_xor(acc, 32);
            if (half == 0) {
                // granule `off / channels`-th block of 576 x channels interleaved samples: element (32 ss + sb) * channels + ch
                const size_t at = (off - ch) * 576 + (size_t)(32 * ss + sb) * channels + ch;
                if (OUT16) a.pcm16[at] = mp3_to_i16(sum);
                else if (a.planar_stride) a.pcm[off * a.planar_stride + (size_t)(32 * ss + sb)] = (float)mp3_to_i16(sum) * (1.0f / 32768.0f);
                else a.pcm[at] = sum;
            }
            wave_sync();
        }
    }
    for (int i = lane; i < 576; i += 64) st[i] = ov[i];
    for (int i = lane; i < 1024; i += 64) st[576 + i] = rg[i];
    if (lane == 0) st[1600] = __uint_as_float(pos);
}

}  // namespace

hipError_t launch_mp3_hybrid(const Mp3Args &a, hipStream_t s) {
    if (a.n_tasks == 0) return hipSuccess;
    const uint32_t blocks = (a.n_tasks + kWaves - 1) / kWaves;
    if (a.pcm16) hipLaunchKernelGGL(k_mp3_hybrid<true>, dim3(blocks), dim3(kWaves * 64), 0, s, a);
    else hipLaunchKernelGGL(k_mp3_hybrid<false>, dim3(blocks), dim3(kWaves * 64), 0, s, a);
    return hipGetLastError();
}

}  // namespace sk

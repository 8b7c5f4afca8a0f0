// pcm.hip -- sample-width / endianness / interleave kernels for gfx950.
//
// Replaces soundkit::audio_bytes (soundkit/src/audio_bytes.rs:3-373), the f32<->PCM helpers of
// soundkit/src/audio_pipeline.rs:17-50 and 74-98, and the conversion ends of the decoder
// worker (soundkit-decoder/src/lib.rs:1815-1827, 3458-3701; soundkit-mp3/src/lib.rs:376-396).
// All of it is HBM-bound byte shuffling: each lane moves 4 samples per step with the widest
// aligned access the element size allows (8/12/16 B loads, 8/16 B stores), integer results
// are bit-exact, and float->int follows Rust's `as` casts (truncate, saturate, NaN -> 0).
#include "sk_device.h"

#include "../../include/soundkit_amd.h"

namespace sk {

namespace {

// ---- Rust cast semantics ---------------------------------------------------------------
__device__ __forceinline__ int f32_as_i32(float x) {
    if (x != x) return 0;
    if (x <= -2147483648.0f) return INT32_MIN;
    if (x >= 2147483648.0f) return INT32_MAX;
    return (int)x;
}
__device__ __forceinline__ int f32_as_i16(float x) {
    if (x != x) return 0;
    if (x <= -32768.0f) return -32768;
    if (x >= 32767.0f) return 32767;
    return (int)x;
}
__device__ __forceinline__ float clamp1(float x) {  // f32::clamp(-1, 1): NaN stays NaN
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    return x;
}
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint32_t bswap16(uint32_t v) { return ((v & 0xff) << 8) | ((v >> 8) & 0xff); }
__device__ __forceinline__ int sext24(uint32_t v) { return (int)(((v & 0xffffffu) ^ 0x800000u) - 0x800000u); }
__device__ __forceinline__ int sext16(uint32_t v) { return (int)(short)(unsigned short)v; }
__device__ __forceinline__ uint32_t be24(uint32_t v) { return ((v & 0xff) << 16) | (v & 0xff00) | ((v >> 16) & 0xff); }

// soundkit-decoder lib.rs:1815-1827; the f64-free exact form of sk_device.h (exhaustively equal, tools/check_f32_rounding.c)
__device__ __forceinline__ int float_sample_to_i16(float s) { return dev_float_sample_to_i16_f32(s); }
__device__ __forceinline__ int mp3_f32_to_i16(float s) {  // soundkit-mp3 lib.rs:376-385
    const float scaled = roundf(s * 32767.0f);
    if (scaled > 32767.0f) return 32767;
    if (scaled < -32768.0f) return -32768;
    return f32_as_i16(scaled);
}
__device__ __forceinline__ int mp3_f32_to_i32(float s) {  // soundkit-mp3 lib.rs:387-396
    const float scaled = roundf(s * 2147483648.0f);
    if (scaled > 2147483648.0f) return INT32_MAX;
    if (scaled < -2147483648.0f) return INT32_MIN;
    return f32_as_i32(scaled);
}
__device__ __forceinline__ int f32_to_i32_pcm(float x) {  // audio_bytes.rs:194-199 (both scales are 2^31 in f32)
    return f32_as_i32(clamp1(x) * 2147483648.0f);
}
__device__ __forceinline__ int f32_to_s24_pcm(float x) {  // audio_bytes.rs:210-216
    const float c = clamp1(x);
    return c >= 0.0f ? f32_as_i32(c * 8388607.0f) : f32_as_i32(c * 8388608.0f);
}

// raw = the input element's bytes, little-endian packed in the low bits; returns output bits
template <int OP>
__device__ __forceinline__ uint32_t convert_raw(uint32_t raw) {
    switch (OP) {
    case SK_PCM_I16LE_TO_F32:
    case SK_PCM_VEC_I16_TO_F32: return __float_as_uint((float)sext16(raw) / 32768.0f);
    case SK_PCM_I16_TO_I16LE:
    case SK_PCM_I16LE_TO_I16:
    case SK_PCM_S16LE_TO_I16:
    case SK_PCM_STEREO_TO_MONO_TAKE_LEFT: return raw & 0xffff;
    case SK_PCM_S24LE_TO_I32: return (uint32_t)sext24(raw);
    case SK_PCM_S24LE_TO_I16: return (uint32_t)(sext24(raw) >> 8) & 0xffff;
    case SK_PCM_S24BE_TO_I16: return (uint32_t)(sext24(be24(raw)) >> 8) & 0xffff;
    case SK_PCM_S32LE_TO_I32: return raw;
    case SK_PCM_S32BE_TO_I32: return bswap32(raw);
    case SK_PCM_S32LE_TO_S24: return raw & 0x00ffffff;
    case SK_PCM_S32BE_TO_S24: return bswap32(raw) & 0x00ffffff;
    case SK_PCM_S32LE_TO_F32:
    case SK_PCM_VEC_I32_TO_F32: return __float_as_uint((float)(int)raw / 2147483648.0f);
    case SK_PCM_S32BE_TO_F32: return __float_as_uint((float)(int)bswap32(raw) / 2147483648.0f);
    case SK_PCM_S32LE_TO_I16: return (uint32_t)((int)raw >> 16) & 0xffff;
    case SK_PCM_S32BE_TO_I16: return (uint32_t)((int)bswap32(raw) >> 16) & 0xffff;
    case SK_PCM_F32LE_TO_I16:
    case SK_PCM_VEC_F32_TO_I16: return (uint32_t)f32_as_i16(clamp1(__uint_as_float(raw)) * 32767.0f) & 0xffff;
    case SK_PCM_F32BE_TO_I16: return (uint32_t)f32_as_i16(clamp1(__uint_as_float(bswap32(raw))) * 32767.0f) & 0xffff;
    case SK_PCM_F32LE_TO_I32: return (uint32_t)f32_to_i32_pcm(__uint_as_float(raw));
    case SK_PCM_F32LE_TO_S24: return (uint32_t)f32_to_s24_pcm(__uint_as_float(raw));
    case SK_PCM_S16BE_TO_I16: return bswap16(raw);
    case SK_PCM_S16LE_TO_I32: return (uint32_t)sext16(raw);
    case SK_PCM_STEREO_TO_MONO_AVG: return (uint32_t)((sext16(raw) + sext16(raw >> 16)) / 2) & 0xffff;
    case SK_PCM_FLOAT_TO_I16_ROUND: return (uint32_t)float_sample_to_i16(__uint_as_float(raw)) & 0xffff;
    case SK_PCM_MP3_F32_TO_I16: return (uint32_t)mp3_f32_to_i16(__uint_as_float(raw)) & 0xffff;
    case SK_PCM_MP3_F32_TO_I32: return (uint32_t)mp3_f32_to_i32(__uint_as_float(raw));
    default: return 0;
    }
}

constexpr int kInBytes[SK_PCM_OP_COUNT] = {2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 2, 2, 4, 4, 4, 2, 4, 4, 4, 4};
constexpr int kOutBytes[SK_PCM_OP_COUNT] = {4, 2, 2, 4, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2, 2, 2, 4, 4, 2, 2, 4, 2, 2, 2, 4, 4, 2, 2, 4};

// element s (0..3) of a group of four IB-byte elements held as IB little-endian dwords
template <int IB>
__device__ __forceinline__ uint32_t extract(const uint32_t (&w)[IB], int s) {
    if (IB == 4) return w[s];
    if (IB == 2) return (w[s >> 1] >> (16 * (s & 1))) & 0xffff;
    // IB == 3: bytes 3s .. 3s+2 of 12
    const int bit = 24 * s;
    const int d = bit >> 5, sh = bit & 31;
    uint64_t pair = (uint64_t)w[d] | ((uint64_t)(d + 1 < IB ? w[d + 1] : 0u) << 32);
    return (uint32_t)(pair >> sh) & 0xffffff;
}

// Scalar element access.  2- and 4-byte elements are naturally aligned (API contract), so they move
// as one typed access; only 3-byte samples go byte by byte.  (Splitting a sign-extended value into
// byte stores is also what hipcc 7.2 folds into a zero-extending v_perm_b32 -- avoid that shape.)
__device__ __forceinline__ uint32_t load_raw_scalar(const uint8_t *p, int ib) {
    if (ib == 4) return *reinterpret_cast<const uint32_t *>(p);
    if (ib == 2) return *reinterpret_cast<const uint16_t *>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
__device__ __forceinline__ void store_raw_scalar(uint8_t *p, uint32_t v, int ob) {
    if (ob == 4) {
        *reinterpret_cast<uint32_t *>(p) = v;
    } else if (ob == 2) {
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)v;
    } else {
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16);
    }
}

// grid-stride over groups of 4 samples; VEC = both buffers 16-byte aligned
template <int OP, bool VEC>
__global__ __launch_bounds__(256) void k_convert(const uint8_t *in, uint8_t *out, size_t n) {
    constexpr int IB = kInBytes[OP], OB = kOutBytes[OP];
    const size_t groups = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        uint32_t w[IB];
        if (VEC) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(in + g * 4 * IB);
            if constexpr (IB == 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src);
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[IB - 1] = v.w;
            } else if constexpr (IB == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(src);
                w[0] = v.x; w[IB - 1] = v.y;
            } else {
#pragma unroll
                for (int d = 0; d < IB; ++d) w[d] = src[d];
            }
        }
        uint32_t r[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t raw = VEC ? extract<IB>(w, s) : load_raw_scalar(in + (g * 4 + s) * IB, IB);
            r[s] = convert_raw<OP>(raw);
        }
        if (VEC) {
            if (OB == 4) {
                *reinterpret_cast<uint4 *>(out + g * 16) = make_uint4(r[0], r[1], r[2], r[3]);
            } else {
                *reinterpret_cast<uint2 *>(out + g * 8) = make_uint2(r[0] | (r[1] << 16), r[2] | (r[3] << 16));
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) store_raw_scalar(out + (g * 4 + s) * OB, r[s], OB);
        }
    }
    // tail (n % 4) by the first few threads of block 0
    const size_t tail0 = groups * 4;
    if (blockIdx.x == 0 && threadIdx.x < n - tail0) {
        const size_t i = tail0 + threadIdx.x;
        store_raw_scalar(out + i * OB, convert_raw<OP>(load_raw_scalar(in + i * IB, IB)), OB);
    }
}

template <int OP>
hipError_t launch_convert_op(const void *in, void *out, size_t n, hipStream_t s) {
    const bool vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    size_t groups = n / 4;
    // one group of four samples per thread, blocks in address order: short-lived waves that each touch one piece of
    // memory stream at 6.3 TB/s on this part, a grid-stride loop over a capped grid at 5.5-5.7 (tools/probe/stream_probe.hip)
    size_t blocks = (groups + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7fffffffu) blocks = 0x7fffffffu;
    if (vec)
        hipLaunchKernelGGL((k_convert<OP, true>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t *)in,
                           (uint8_t *)out, n);
    else
        hipLaunchKernelGGL((k_convert<OP, false>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t *)in,
                           (uint8_t *)out, n);
    return hipGetLastError();
}

// ---- layout kernels ------------------------------------------------------------------------

// interleaved [frames][ch] <-> planar [ch][frames], element = EB bytes moved as-is
template <typename T, bool TO_PLANAR>
__global__ __launch_bounds__(256) void k_transpose(const T *in, T *out, size_t frames, uint32_t ch) {
    const size_t total = frames * ch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        // i walks the interleaved side: frame = i / ch, channel = i % ch
        const size_t f = i / ch;
        const uint32_t c = (uint32_t)(i - f * ch);
        if (TO_PLANAR)
            out[(size_t)c * frames + f] = in[i];
        else
            out[i] = in[(size_t)c * frames + f];
    }
}

__global__ __launch_bounds__(256) void k_deinterleave_s24(const uint8_t *in, int32_t *planar, size_t frames, uint32_t ch) {
    const size_t total = frames * ch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t f = i / ch;
        const uint32_t c = (uint32_t)(i - f * ch);
        planar[(size_t)c * frames + f] = sext24(load_raw_scalar(in + i * 3, 3));
    }
}

__device__ __forceinline__ float sample_to_f32(int variant, int fmt, uint32_t raw) {
    float s;
    switch (fmt) {
    case SK_FMT_F32LE: s = __uint_as_float(raw); break;
    case SK_FMT_F32BE: s = __uint_as_float(bswap32(raw)); break;
    case SK_FMT_S16LE: s = (float)sext16(raw) / 32768.0f; break;
    case SK_FMT_S16BE: s = (float)sext16(bswap16(raw)) / 32768.0f; break;
    case SK_FMT_S24LE: s = (float)sext24(raw) / (variant == 0 ? 8388608.0f : 2147483648.0f); break;
    case SK_FMT_S24BE: s = (float)sext24(be24(raw)) / 8388608.0f; break;
    case SK_FMT_S32LE: s = (float)(int)raw / 2147483648.0f; break;
    default: s = (float)(int)bswap32(raw) / 2147483648.0f; break;
    }
    if (variant == 0 && !isfinite(s)) s = 0.0f;  // soundkit-decoder lib.rs:3614
    return s;
}

__global__ __launch_bounds__(256) void k_bytes_to_f32_planar(int variant, int fmt, int bps, const uint8_t *in,
                                                             size_t frames, uint32_t ch, float *planar) {
    const size_t total = frames * ch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t f = i / ch;
        const uint32_t c = (uint32_t)(i - f * ch);
        uint32_t raw;
        if (bps == 2) raw = reinterpret_cast<const uint16_t *>(in)[i];
        else if (bps == 4) raw = reinterpret_cast<const uint32_t *>(in)[i];
        else raw = load_raw_scalar(in + i * 3, 3);
        planar[(size_t)c * frames + f] = sample_to_f32(variant, fmt, raw);
    }
}

// stereo s16le fast path: 4 frames per lane, 16 B in, 2 x 16 B out
__global__ __launch_bounds__(256) void k_s16le_stereo_to_f32_planar(const uint8_t *in, size_t frames, float *planar) {
    const size_t groups = frames / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(in)[g];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float l[4], r[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            l[s] = (float)sext16(w[s]) / 32768.0f;
            r[s] = (float)sext16(w[s] >> 16) / 32768.0f;
        }
        reinterpret_cast<float4 *>(planar)[g] = make_float4(l[0], l[1], l[2], l[3]);
        reinterpret_cast<float4 *>(planar + frames)[g] = make_float4(r[0], r[1], r[2], r[3]);
    }
    for (size_t f = groups * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < frames; f += stride) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(in)[f];
        planar[f] = (float)sext16(w) / 32768.0f;
        planar[frames + f] = (float)sext16(w >> 16) / 32768.0f;
    }
}

__device__ __forceinline__ uint32_t f32_to_sample(int fmt, float x) {
    switch (fmt) {
    case SK_FMT_F32LE: return __float_as_uint(x);
    case SK_FMT_S16LE: return (uint32_t)float_sample_to_i16(x) & 0xffff;
    case SK_FMT_S24LE: {  // soundkit-decoder lib.rs:3649-3661
        const float c = clamp1(x);
        return (uint32_t)(c >= 0.0f ? f32_as_i32(c * 8388607.0f) : f32_as_i32(c * 8388608.0f)) & 0xffffff;
    }
    default: return (uint32_t)f32_to_i32_pcm(x);  // S32LE, lib.rs:3664-3677
    }
}

__global__ __launch_bounds__(256) void k_f32_planar_to_bytes(int fmt, int bps, const float *planar, size_t frames,
                                                             uint32_t ch, uint8_t *out) {
    const size_t total = frames * ch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t f = i / ch;
        const uint32_t c = (uint32_t)(i - f * ch);
        const uint32_t v = f32_to_sample(fmt, planar[(size_t)c * frames + f]);
        if (bps == 2) reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)v;
        else if (bps == 4) reinterpret_cast<uint32_t *>(out)[i] = v;
        else store_raw_scalar(out + i * 3, v, 3);
    }
}

// stereo planar f32 -> interleaved s16le fast path (the worker's common output): 4 frames per lane
__global__ __launch_bounds__(256) void k_f32_planar_stereo_to_s16le(const float *planar, size_t frames, uint8_t *out) {
    const size_t groups = frames / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const float4 l = reinterpret_cast<const float4 *>(planar)[g];
        const float4 r = reinterpret_cast<const float4 *>(planar + frames)[g];
        uint4 v;
        v.x = ((uint32_t)float_sample_to_i16(l.x) & 0xffff) | ((uint32_t)float_sample_to_i16(r.x) << 16);
        v.y = ((uint32_t)float_sample_to_i16(l.y) & 0xffff) | ((uint32_t)float_sample_to_i16(r.y) << 16);
        v.z = ((uint32_t)float_sample_to_i16(l.z) & 0xffff) | ((uint32_t)float_sample_to_i16(r.z) << 16);
        v.w = ((uint32_t)float_sample_to_i16(l.w) & 0xffff) | ((uint32_t)float_sample_to_i16(r.w) << 16);
        reinterpret_cast<uint4 *>(out)[g] = v;
    }
    for (size_t f = groups * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < frames; f += stride) {
        reinterpret_cast<uint32_t *>(out)[f] = ((uint32_t)float_sample_to_i16(planar[f]) & 0xffff) |
                                               ((uint32_t)float_sample_to_i16(planar[frames + f]) << 16);
    }
}

// batch of [2][plane_stride] planar f32 -> [frames][2] interleaved s16le.  The output of all batches is
// one contiguous array of (L, R) dwords; a lane owns 4 consecutive dwords of it (a 16-byte aligned
// store whatever `frames` is) and fetches its 8 samples with coalesced dword loads.
__global__ __launch_bounds__(256) void k_f32_planar_stereo_to_s16le_batch(const float *planar, size_t plane_stride,
                                                                          size_t frames, size_t batch, uint8_t *out) {
    const size_t total = batch * frames;
    const size_t groups = (total + 3) / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const size_t g0 = g * 4;
        size_t b = g0 / frames, f = g0 - b * frames;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t w = 0;
            if (g0 + k < total) {
                const float *src = planar + b * 2 * plane_stride + f;
                w = ((uint32_t)float_sample_to_i16(src[0]) & 0xffff) | ((uint32_t)float_sample_to_i16(src[plane_stride]) << 16);
            }
            v[k] = w;
            if (++f == frames) { f = 0; ++b; }
        }
        if (g0 + 3 < total) {
            reinterpret_cast<uint4 *>(out)[g] = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (g0 + k < total) reinterpret_cast<uint32_t *>(out)[g0 + k] = v[k];
        }
    }
}

__global__ __launch_bounds__(256) void k_f32_planar_to_bytes_batch(int fmt, int bps, const float *planar,
                                                                   size_t plane_stride, size_t frames, uint32_t ch,
                                                                   uint8_t *out) {
    const float *src = planar + (size_t)blockIdx.y * ch * plane_stride;
    uint8_t *dst = out + (size_t)blockIdx.y * frames * ch * bps;
    const size_t total = frames * ch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t f = i / ch;
        const uint32_t c = (uint32_t)(i - f * ch);
        const uint32_t v = f32_to_sample(fmt, src[(size_t)c * plane_stride + f]);
        if (bps == 2) reinterpret_cast<uint16_t *>(dst)[i] = (uint16_t)v;
        else if (bps == 4) reinterpret_cast<uint32_t *>(dst)[i] = v;
        else store_raw_scalar(dst + i * 3, v, 3);
    }
}

__global__ __launch_bounds__(256) void k_downmix_mono(const float *planar, size_t frames, uint32_t ch, float *mono) {
    const float scale = 1.0f / (float)ch;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += stride) {
        float acc = 0.0f;  // lib.rs:3500-3508: mono[i] += sample * scale, channels in order
        for (uint32_t c = 0; c < ch; ++c) acc += planar[(size_t)c * frames + i] * scale;
        mono[i] = acc;
    }
}

__global__ __launch_bounds__(256) void k_exact_to_i16(int fmt, const uint8_t *in, size_t samples, uint8_t *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < samples; i += stride) {
        int s;
        switch (fmt) {
        case SK_FMT_S24LE: s = sext24(load_raw_scalar(in + i * 3, 3)) >> 8; break;
        case SK_FMT_S24BE: s = sext24(be24(load_raw_scalar(in + i * 3, 3))) >> 8; break;
        case SK_FMT_S32LE: s = (int)reinterpret_cast<const uint32_t *>(in)[i] >> 16; break;
        default: s = (int)bswap32(reinterpret_cast<const uint32_t *>(in)[i]) >> 16; break;
        }
        reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)s;
    }
}

inline unsigned grid_for(size_t items) {
    size_t blocks = (items + 255) / 256;  // one item per thread (the kernels keep their grid-stride loops for larger inputs)
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7fffffffu) blocks = 0x7fffffffu;
    return (unsigned)blocks;
}

// one workgroup = 1024 frames of one job
__global__ __launch_bounds__(256) void k_pack_jobs(const PackJob *jobs, uint32_t n_jobs) {
    const uint32_t j = blockIdx.y;
    if (j >= n_jobs) return;
    const PackJob job = jobs[j];
    const uint32_t f0 = blockIdx.x * 1024;
    if (f0 >= job.frames) return;
    const uint32_t f1 = min(f0 + 1024u, job.frames);
    const int fmt = job.bits == 16 ? SK_FMT_S16LE : (job.bits == 24 ? SK_FMT_S24LE : SK_FMT_S32LE);
    const uint32_t bps = job.bits / 8;
    for (uint32_t f = f0 + threadIdx.x; f < f1; f += 256) {
        float x[2];
        x[0] = job.src0[f];
        x[1] = job.ch_in > 1 ? job.src1[f] : 0.0f;
        if (job.mode == kPackFromQ && job.bits == 16 && job.ch_out == job.ch_in) {  // the fast path on rows that hold q / 32768
            if (job.ch_in > 1) {
                const uint32_t lo = (uint32_t)(int)(x[0] * 32768.0f) & 0xffff, hi = (uint32_t)(int)(x[1] * 32768.0f) & 0xffff;
                reinterpret_cast<uint32_t *>(job.dst)[f] = lo | (hi << 16);
            } else {
                reinterpret_cast<uint16_t *>(job.dst)[f] = (uint16_t)(int)(x[0] * 32768.0f);
            }
            continue;
        }
        if (job.mode == kPackDirect) {
            if (job.ch_in > 1) {
                const uint32_t lo = (uint32_t)float_sample_to_i16(x[0]) & 0xffff, hi = (uint32_t)float_sample_to_i16(x[1]) & 0xffff;
                reinterpret_cast<uint32_t *>(job.dst)[f] = lo | (hi << 16);
            } else {
                reinterpret_cast<uint16_t *>(job.dst)[f] = (uint16_t)float_sample_to_i16(x[0]);
            }
            continue;
        }
        if (job.mode == kPackViaS16)
            for (int c = 0; c < 2; ++c) x[c] = (float)float_sample_to_i16(x[c]) / 32768.0f;
        uint32_t ch = job.ch_in;
        if (job.ch_out < job.ch_in) {  // downmix_channels(.., 1): mono += sample * (1 / channels), channels in order
            const float scale = 1.0f / (float)job.ch_in;
            float acc = 0.0f;
            for (uint32_t c = 0; c < job.ch_in; ++c) acc += x[c] * scale;
            x[0] = acc;
            ch = 1;
        }
        for (uint32_t c = 0; c < ch; ++c) {
            const uint32_t v = f32_to_sample(fmt, x[c]);
            const size_t i = (size_t)f * ch + c;
            if (bps == 2) reinterpret_cast<uint16_t *>(job.dst)[i] = (uint16_t)v;
            else if (bps == 4) reinterpret_cast<uint32_t *>(job.dst)[i] = v;
            else store_raw_scalar(job.dst + i * 3, v, 3);
        }
    }
}

}  // namespace

hipError_t launch_pack_jobs(const PackJob *jobs, uint32_t n_jobs, uint32_t max_frames, hipStream_t s) {
    if (n_jobs == 0 || max_frames == 0) return hipSuccess;
    for (uint32_t j0 = 0; j0 < n_jobs; j0 += 65535) {
        const uint32_t n = n_jobs - j0 < 65535 ? n_jobs - j0 : 65535;
        hipLaunchKernelGGL(k_pack_jobs, dim3((max_frames + 1023) / 1024, n), dim3(256), 0, s, jobs + j0, n);
    }
    return hipGetLastError();
}

hipError_t launch_pcm_convert(int op, const void *in, void *out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    switch (op) {
#define SK_CASE(OPV) case OPV: return launch_convert_op<OPV>(in, out, n, s);
        SK_CASE(0) SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8) SK_CASE(9)
        SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16) SK_CASE(17) SK_CASE(18)
        SK_CASE(19) SK_CASE(20) SK_CASE(21) SK_CASE(22) SK_CASE(23) SK_CASE(24) SK_CASE(25) SK_CASE(26) SK_CASE(27)
        SK_CASE(28)
#undef SK_CASE
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_interleave(const void *planar, void *out, size_t frames, uint32_t ch, int eb, hipStream_t s) {
    if (frames == 0 || ch == 0) return hipSuccess;
    const unsigned g = grid_for(frames * ch);
    if (eb == 2)
        hipLaunchKernelGGL((k_transpose<uint16_t, false>), dim3(g), dim3(256), 0, s, (const uint16_t *)planar,
                           (uint16_t *)out, frames, ch);
    else
        hipLaunchKernelGGL((k_transpose<uint32_t, false>), dim3(g), dim3(256), 0, s, (const uint32_t *)planar,
                           (uint32_t *)out, frames, ch);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const void *in, void *planar, size_t frames, uint32_t ch, int eb, hipStream_t s) {
    if (frames == 0 || ch == 0) return hipSuccess;
    const unsigned g = grid_for(frames * ch);
    if (eb == 2)
        hipLaunchKernelGGL((k_transpose<uint16_t, true>), dim3(g), dim3(256), 0, s, (const uint16_t *)in,
                           (uint16_t *)planar, frames, ch);
    else
        hipLaunchKernelGGL((k_transpose<uint32_t, true>), dim3(g), dim3(256), 0, s, (const uint32_t *)in,
                           (uint32_t *)planar, frames, ch);
    return hipGetLastError();
}

hipError_t launch_deinterleave_s24(const uint8_t *in, int32_t *planar, size_t frames, uint32_t ch, hipStream_t s) {
    if (frames == 0 || ch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_deinterleave_s24, dim3(grid_for(frames * ch)), dim3(256), 0, s, in, planar, frames, ch);
    return hipGetLastError();
}

hipError_t launch_bytes_to_f32_planar(int variant, int fmt, const uint8_t *in, size_t frames, uint32_t ch, float *planar,
                                      hipStream_t s) {
    if (frames == 0 || ch == 0) return hipSuccess;
    const int bps = sk_pcm_fmt_bytes(fmt);
    if (fmt == SK_FMT_S16LE && ch == 2 && ((((uintptr_t)in | (uintptr_t)planar) & 15) == 0) && frames % 4 == 0) {
        hipLaunchKernelGGL(k_s16le_stereo_to_f32_planar, dim3(grid_for(frames / 4)), dim3(256), 0, s, in, frames, planar);
    } else {
        hipLaunchKernelGGL(k_bytes_to_f32_planar, dim3(grid_for(frames * ch)), dim3(256), 0, s, variant, fmt, bps, in,
                           frames, ch, planar);
    }
    return hipGetLastError();
}

hipError_t launch_f32_planar_to_bytes(int fmt, const float *planar, size_t frames, uint32_t ch, uint8_t *out,
                                      hipStream_t s) {
    if (frames == 0 || ch == 0) return hipSuccess;
    const int bps = sk_pcm_fmt_bytes(fmt);
    if (fmt == SK_FMT_S16LE && ch == 2 && ((((uintptr_t)out | (uintptr_t)planar) & 15) == 0) && frames % 4 == 0) {
        hipLaunchKernelGGL(k_f32_planar_stereo_to_s16le, dim3(grid_for(frames / 4)), dim3(256), 0, s, planar, frames, out);
    } else {
        hipLaunchKernelGGL(k_f32_planar_to_bytes, dim3(grid_for(frames * ch)), dim3(256), 0, s, fmt, bps, planar, frames,
                           ch, out);
    }
    return hipGetLastError();
}

hipError_t launch_f32_planar_to_bytes_batch(int fmt, const float *planar, size_t batch, size_t plane_stride, size_t frames,
                                            uint32_t ch, uint8_t *out, hipStream_t s) {
    if (batch == 0 || frames == 0 || ch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    const int bps = sk_pcm_fmt_bytes(fmt);
    if (fmt == SK_FMT_S16LE && ch == 2 && (((uintptr_t)out & 15) == 0)) {
        size_t blocks = ((batch * frames + 3) / 4 + 255) / 256;
        if (blocks > 0x7fffffffu) blocks = 0x7fffffffu;  // one group per thread, blocks in address order (see launch_convert_op)
        hipLaunchKernelGGL(k_f32_planar_stereo_to_s16le_batch, dim3((unsigned)blocks), dim3(256), 0, s, planar,
                           plane_stride, frames, batch, out);
    } else {
        unsigned gx = (unsigned)((frames * ch + 255) / 256);
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(k_f32_planar_to_bytes_batch, dim3(gx, (unsigned)batch), dim3(256), 0, s, fmt, bps, planar,
                           plane_stride, frames, ch, out);
    }
    return hipGetLastError();
}

hipError_t launch_downmix_mono(const float *planar, size_t frames, uint32_t ch, float *mono, hipStream_t s) {
    if (frames == 0 || ch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_downmix_mono, dim3(grid_for(frames)), dim3(256), 0, s, planar, frames, ch, mono);
    return hipGetLastError();
}

hipError_t launch_exact_to_i16(int fmt, const uint8_t *in, size_t samples, uint8_t *out, hipStream_t s) {
    if (samples == 0) return hipSuccess;
    hipLaunchKernelGGL(k_exact_to_i16, dim3(grid_for(samples)), dim3(256), 0, s, fmt, in, samples, out);
    return hipGetLastError();
}

}  // namespace sk

// aac_entropy.hip -- the AAC-LC access-unit front-end on gfx950, one access unit per lane (SURVEY 8f ranks 1 + 4).
//
// aac_entropy_core.h -- the same source that tests/entropy_core_check.cpp proves equal to the host front-end under
// AddressSanitizer -- decodes the units and writes the dequantised, stereo- and TNS-processed spectra straight into the
// synthesis kernel's input buffer, plus the window fields into the synthesis schedule.  A unit that fails gets its
// status recorded, zero spectra and a plain long window (so the synthesis launch that follows needs no special case),
// and ends its stream for this launch: later units of the stream are marked skipped.
#include "aac_entropy_core.h"
#include "sk_device.h"


namespace sk {

namespace {

// ---- frame-parallel form ---------------------------------------------------------------------------------------------
// The only thing that orders the access units of a stream is the PNS generator.  Phase 1 decodes every unit in its own
// lane with the noise bands left open (and counted), phase 2 walks each stream once to hand every unit its generator
// state by LCG jump-ahead, phase 3 fills the noise and finishes (stereo tools, TNS, rest of the unit) per unit again.
// The composition equals decode_access_unit: tests/entropy_core_check.cpp checks that on the CPU for every unit it sees.

__device__ __forceinline__ sk_ec::Tables lds_tables(const EntropyArgs &a, uint4 *lds_raw) {
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds_raw);
    for (uint32_t i = threadIdx.x; i < a.lds_bytes / 16; i += blockDim.x)
        lds_raw[i] = reinterpret_cast<const uint4 *>(a.lds_blob)[i];
    __syncthreads();
    sk_ec::Tables t = a.t;
    t.meta = reinterpret_cast<const uint32_t *>(lds + a.lds_meta_off);
    t.lut = reinterpret_cast<const uint32_t *>(lds + a.lds_lut_off);
    t.tuples = reinterpret_cast<const uint64_t *>(lds + a.lds_tuple_off);
    t.sf_mult = reinterpret_cast<const float *>(lds + a.lds_sf_off);
    t.swb = reinterpret_cast<const uint16_t *>(lds + a.lds_swb_off);
    t.pow43_lo = reinterpret_cast<const float *>(lds + a.lds_pow_off);
    return t;
}

// Lanes per wave that carry a unit: a wave executes the union of its lanes' paths (codebooks, sign and escape branches
// differ from unit to unit), and a tick of 65 536 units on 64 lanes per wave is one wave per SIMD, every latency in
// the open.  Fewer units per wave (the first 64 >> lane_shift lanes) means shorter unions and more waves to hide
// latency behind, for more wave-instructions in total on SIMDs that were mostly waiting (profiles/r01_pmc_entropy.md).
__device__ __forceinline__ bool unit_of_lane(const EntropyArgs &a, uint32_t &k) {
    const uint32_t per_wave = 64u >> a.lane_shift;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = wave * per_wave + lane;
    const bool mine = lane < per_wave && slot < a.n_units;
    k = (mine && a.order) ? a.order[slot] : slot;
    return mine;
}

// The spectra of a wave's units, zeroed by all 64 lanes together (16 bytes per lane, whole cache lines) before any lane
// decodes: left to the lanes it is 1024 stores per channel, each a single word per lane 8 KiB from its neighbour's.
__device__ __forceinline__ void wave_zero_spectra(const EntropyArgs &a) {
    const uint32_t per_wave = 64u >> a.lane_shift;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    for (uint32_t l = 0; l < per_wave; ++l) {
        const uint32_t slot = wave * per_wave + l;  // wave-uniform
        if (slot >= a.n_units) break;
        const EntropyUnit u = a.units[a.order ? a.order[slot] : slot];
        const uint32_t quads = a.tasks[u.task].channels * 256u;
        float4 *dst = reinterpret_cast<float4 *>(a.coeffs + (size_t)u.off1024 * 1024);
        for (uint32_t i = lane; i < quads; i += 64) {
            dst[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    // a lane's own stores to these addresses come later in the same wave's store stream; wait all the same
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void k_aac_entropy_parse(EntropyArgs a) {
    extern __shared__ uint4 lds_raw[];
    const sk_ec::Tables t = lds_tables(a, lds_raw);
    wave_zero_spectra(a);
    uint32_t k;
    if (!unit_of_lane(a, k)) return;
    const EntropyUnit u = a.units[k];
    const EntropyTask tk = a.tasks[u.task];
    sk_ec::Stream st{tk.sf_index, (int)tk.channels, 0u, true};
    sk_ec::Scratch side;
    uint8_t seq[2] = {0, 0}, shape[2] = {0, 0};
    const int status = sk_ec::parse_unit(t, st, a.words + u.word_offset, u.byte_len, a.coeffs + (size_t)u.off1024 * 1024, seq, shape, side,
                                         sk_ec::PNS_COUNT);
    a.status[k] = status;
    a.side[k] = side;
    for (uint32_t c = 0; c < tk.channels; ++c) a.entries[u.entry[c]].win = (uint32_t)seq[c] | ((uint32_t)shape[c] << 2);
}

// Phase one of the quantised hand-over: the host has done the Huffman decode; a lane rebuilds its unit's side
// information from the wire record, dequantises the integers into the synthesis input and leaves the noise bands open.
__global__ __launch_bounds__(512) void k_aac_expand_q(EntropyArgs a) {
    extern __shared__ uint4 lds_raw[];
    const sk_ec::Tables t = lds_tables(a, lds_raw);
    uint32_t k;
    if (!unit_of_lane(a, k)) return;
    const EntropyUnit u = a.units[k];
    const EntropyTask tk = a.tasks[u.task];
    const sk_ec::Stream st{tk.sf_index, (int)tk.channels, 0u};
    // every byte of the record the later phases read is defined: link / finish / seal take it from global memory, and
    // sk_tick_run_q is a public entry point (a caller's record is checked, not trusted)
    sk_ec::Scratch side{};
    int status = sk_ec::unpack_unit(t, a.wire[k], side);
    if (status == sk_ec::EC_OK && (uint32_t)a.wire[k].channels != tk.channels) status = sk_ec::EC_INVALID_CONFIG;
    float *coef = a.coeffs + (size_t)u.off1024 * 1024;
    const int16_t *q = a.quant + (size_t)u.off1024 * 1024;
    for (uint32_t c = 0; c < tk.channels && status == sk_ec::EC_OK; ++c) status = sk_ec::dequant_channel(t, st, side.ch[c], q + 1024 * c, coef + 1024 * c);
    if (status == sk_ec::EC_OK) status = sk_ec::apply_wide(a.wire[k], tk.channels, side, t, st, q, coef);
    if (status != sk_ec::EC_OK) side.noise_samples = 0;  // a record that was rejected consumed no noise
    a.status[k] = status;
    a.side[k] = side;
    for (uint32_t c = 0; c < tk.channels; ++c)
        a.entries[u.entry[c]].win = status == sk_ec::EC_OK ? ((uint32_t)side.ch[c].ics.sequence | ((uint32_t)side.ch[c].ics.shape << 2)) : 0u;
}

__global__ __launch_bounds__(64) void k_aac_entropy_link(EntropyArgs a) {
    const uint32_t task = blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= a.n_tasks) return;
    const EntropyTask tk = a.tasks[task];
    uint32_t state = a.pns_state[tk.stream];
    bool dead = false;
    for (uint32_t k = tk.first; k < tk.first + tk.count; ++k) {
        if (dead) {
            a.status[k] = EC_SKIPPED;
            continue;
        }
        a.pns_start[k] = state;
        // a unit that failed while parsing still consumed the noise it had counted up to that point (the sequential
        // decoder's generator has advanced that far when it returns the error)
        state = sk_ec::pns_advance(state, a.side[k].noise_samples);
        if (a.status[k] != sk_ec::EC_OK) dead = true;
    }
    a.pns_state[tk.stream] = state;
}

// After the third phase: a unit can also fail there (stereo tools, TNS, trailing bits, an all-zero noise band), when the
// units behind it in its stream have long been linked and finished.  The sequential decoder would not have touched
// them: mark them skipped, silence them, and put the stream's generator where the failed unit left it (all of its
// noise is generated before any of those checks).  One lane per stream; streams without such a failure only read
// their statuses.
__global__ __launch_bounds__(64) void k_aac_entropy_seal(EntropyArgs a) {
    const uint32_t task = blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= a.n_tasks) return;
    const EntropyTask tk = a.tasks[task];
    uint32_t k = tk.first;
    const uint32_t end = tk.first + tk.count;
    while (k < end && a.status[k] == sk_ec::EC_OK) ++k;
    if (k >= end || k + 1 >= end || a.status[k + 1] == EC_SKIPPED) return;  // no failure, or one the link phase already handled
    a.pns_state[tk.stream] = sk_ec::pns_advance(a.pns_start[k], a.side[k].noise_samples);
    for (uint32_t j = k + 1; j < end; ++j) {
        const EntropyUnit u = a.units[j];
        a.status[j] = EC_SKIPPED;
        float *coef = a.coeffs + (size_t)u.off1024 * 1024;
        for (uint32_t i = 0; i < tk.channels * 1024u; ++i) coef[i] = 0.0f;
        for (uint32_t c = 0; c < tk.channels; ++c) a.entries[u.entry[c]].win = 0;
    }
}

__global__ __launch_bounds__(512) void k_aac_entropy_finish(EntropyArgs a) {
    extern __shared__ uint4 lds_raw[];
    const sk_ec::Tables t = lds_tables(a, lds_raw);
    uint32_t k;
    if (!unit_of_lane(a, k)) return;
    const EntropyUnit u = a.units[k];
    const EntropyTask tk = a.tasks[u.task];
    float *coef = a.coeffs + (size_t)u.off1024 * 1024;
    int status = a.status[k];
    if (status == sk_ec::EC_OK) {
        sk_ec::Stream st{tk.sf_index, (int)tk.channels, a.pns_start[k]};
        // the side record is read where parse left it (global memory): a private copy is 3.1 KB per lane through scratch
        status = sk_ec::finish_unit(t, st, a.words + u.word_offset, u.byte_len, coef, a.side[k], true);
        if (status == sk_ec::EC_OK && a.wire) status = a.wire[k].tail_status;  // the host has looked at the rest of the unit
        a.status[k] = status;
    }
    if (status != sk_ec::EC_OK) {  // failed or skipped: silence for the synthesis launch that follows
        for (uint32_t i = 0; i < tk.channels * 1024u; ++i) coef[i] = 0.0f;
        for (uint32_t c = 0; c < tk.channels; ++c) a.entries[u.entry[c]].win = 0;
    }
}

}  // namespace

hipError_t launch_aac_entropy_parallel(const EntropyArgs &a, hipStream_t s) {
    if (a.n_units == 0) return hipSuccess;
    if (a.lane_shift > 4) return hipErrorInvalidValue;
    // workgroups of eight waves share one copy of the tables in LDS (~60 KB): two per CU, four waves per SIMD
    const uint32_t per_wave = 64u >> a.lane_shift, waves = (a.n_units + per_wave - 1) / per_wave, blocks = (waves + 7) / 8;
    if (a.wire) hipLaunchKernelGGL(k_aac_expand_q, dim3(blocks), dim3(512), a.lds_bytes, s, a);
    else hipLaunchKernelGGL(k_aac_entropy_parse, dim3(blocks), dim3(512), a.lds_bytes, s, a);
    hipLaunchKernelGGL(k_aac_entropy_link, dim3((a.n_tasks + 63) / 64), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_aac_entropy_finish, dim3(blocks), dim3(512), a.lds_bytes, s, a);
    hipLaunchKernelGGL(k_aac_entropy_seal, dim3((a.n_tasks + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace sk

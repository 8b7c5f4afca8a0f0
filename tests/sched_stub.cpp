// CPU harness for the batch scheduler (csrc/pipeline.cpp) with a stand-in engine, built with ThreadSanitizer by
// tests/test_scheduler_cpu.py.  The real AAC front-end parses the real fixture; only the device side is replaced:
// sk_tick_run here emits, per access unit, an "AudioData" whose bytes carry the stream id, the access unit's
// running number within the stream and a checksum of its spectra, so ordering, completeness and isolation can be
// checked without a GPU.  (Test infrastructure: the product library never contains this file.)
#include "../soundkit_amd/csrc/pipeline.cpp"
#include "../soundkit_amd/csrc/aac_frontend.cpp"
#include "../soundkit_amd/csrc/mp3_bitstream.cpp"  // MP3 streams: the real framing, reservoir and Huffman stage (the standard's tables)
#include "../soundkit_amd/csrc/mp3_decoder.cpp"
#include "../soundkit_amd/csrc/load_gen.cpp"  // the bench's load generator: its feeder / consumer loops run under the sanitizers too

#include <atomic>
#include <cstdio>
#include <map>

// ---- stand-ins for the HIP runtime and the engine -------------------------------------------------------
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
}

struct sk_engine {
    std::mutex mu;
    std::vector<uint8_t> open, channels;
    std::vector<uint32_t> next_unit;
    std::vector<uint32_t> rs_fill;  // resampling streams: access units waiting for their chunk of four (lib.rs:1970-2003)
    std::atomic<int> fail_ticks{0}; // the next n ticks fail (a launch failure in the real engine)
};
// the slow-tick regime: every tick holds the engine for this long (the device as the slowest stage)
static std::atomic<uint32_t> g_tick_delay_us{0};
static void tick_delay() {
    const uint32_t us = g_tick_delay_us.load();
    if (us) std::this_thread::sleep_for(std::chrono::microseconds(us));
}
extern "C" {
int sk_engine_device(const sk_engine *) { return 0; }
const char *sk_engine_where(const sk_engine *) { return "stub"; }
uint32_t sk_engine_max_streams(const sk_engine *e) { return (uint32_t)e->open.size(); }
int sk_engine_create(int, uint32_t max_streams, sk_engine **out) {  // lanes 1.. of a pipeline make their own
    sk_engine *e = new sk_engine();
    e->open.assign(max_streams, 0);
    e->channels.assign(max_streams, 0);
    e->next_unit.assign(max_streams, 0);
    e->rs_fill.assign(max_streams, 0);
    *out = e;
    return SK_OK;
}
void sk_engine_destroy(sk_engine *e) { delete e; }
const char *sk_strerror(int) { return "stub"; }
int sk_stream_open(sk_engine *e, uint32_t, uint8_t ch, uint32_t *out) {
    std::lock_guard<std::mutex> lk(e->mu);
    for (uint32_t i = 0; i < e->open.size(); ++i)
        if (!e->open[i]) {
            e->open[i] = 1;
            e->channels[i] = ch;
            e->next_unit[i] = 0;
            if (e->rs_fill.size() < e->open.size()) e->rs_fill.assign(e->open.size(), 0);
            e->rs_fill[i] = 0;
            *out = i;
            return SK_OK;
        }
    return SK_ERR_CAPACITY;
}
int sk_stream_close(sk_engine *e, uint32_t id) {
    std::lock_guard<std::mutex> lk(e->mu);
    if (id >= e->open.size() || !e->open[id]) return SK_ERR_BAD_STREAM;
    e->open[id] = 0;
    return SK_OK;
}
int sk_resampler_open(sk_engine *, uint32_t, uint32_t, uint32_t) { return SK_OK; }
size_t sk_tick_out_bound(const sk_tick_stream *ts, uint32_t n, uint32_t *max_outputs) {
    size_t bytes = 0;
    uint32_t outs = 0;
    for (uint32_t i = 0; i < n; ++i) {
        bytes += (size_t)ts[i].n_frames * 64 + 64;
        outs += ts[i].n_frames + 1;
    }
    *max_outputs = outs;
    return bytes;
}
size_t sk_tick_out_bound_on(sk_engine *, const sk_tick_stream *ts, uint32_t n, uint32_t *max_outputs) { return sk_tick_out_bound(ts, n, max_outputs); }
// What every stand-in tick shares: one "AudioData" per access unit -- or, for a resampling stream, per chunk of four
// units plus the flush tail, the boundaries of the real tick (lib.rs:1970-2003) -- carrying (stream, running unit
// number, checksum, magic).
struct Emit {
    sk_engine *e;
    uint8_t *out;
    size_t out_cap;
    sk_tick_output *outs;
    uint32_t outs_cap;
    uint32_t k = 0;
    size_t cursor = 0;
    int put(const sk_tick_stream &t, uint32_t row, uint32_t frames, uint32_t sum) {
        if (k >= outs_cap || cursor + 64 > out_cap) return SK_ERR_INVALID_ARG;
        uint32_t words[4] = {t.stream, e->next_unit[t.stream], sum, 0xabcd1234u};
        std::memcpy(out + cursor, words, 16);
        outs[k++] = sk_tick_output{row, frames, cursor, 16, 0, e->channels[t.stream], 16, 0};
        cursor += 64;
        return SK_OK;
    }
    int unit(const sk_tick_stream &t, uint32_t row, uint32_t sum, uint32_t frames = 1024) {
        int rc = SK_OK;
        if (!t.resample) rc = put(t, row, frames, sum);
        else if (++e->rs_fill[t.stream] == 4) {
            e->rs_fill[t.stream] = 0;
            rc = put(t, row, 4096, sum);
        }
        ++e->next_unit[t.stream];
        return rc;
    }
    int end_of_row(const sk_tick_stream &t, uint32_t row) {
        if (!t.resample || !t.flush || !e->rs_fill[t.stream]) return SK_OK;
        const uint32_t left = e->rs_fill[t.stream];
        e->rs_fill[t.stream] = 0;
        return put(t, row, 1024 * left, 0);
    }
};
#define STUB_TICK_PROLOGUE()                                                             \
    std::lock_guard<std::mutex> lk(e->mu);                                               \
    tick_delay();                                                                        \
    if (e->fail_ticks.load() > 0 && e->fail_ticks.fetch_sub(1) > 0) return SK_ERR_HIP; \
    std::map<uint32_t, int> seen;                                                        \
    Emit em{e, out, out_cap, outs, outs_cap}

// what pipeline.cpp / mp3_decoder.cpp call for MP3 on the engine
int sk_mp3_set_band_tables(sk_engine *, uint32_t, const uint16_t *, const uint16_t *, const uint8_t *) { return SK_OK; }
int sk_mp3_set_synthesis_window(sk_engine *, const float *) { return SK_OK; }
int sk_mp3_decode_granules_f32(sk_engine *, const sk_mp3_requant_granule *, const sk_mp3_granule_desc *, const int16_t *, float *, uint32_t, int32_t *) {
    return SK_ERR_UNSUPPORTED;  // the decoder handle is not what the scheduler drives
}
int sk_mp3_decode_granules_s16(sk_engine *, const sk_mp3_requant_granule *, const sk_mp3_granule_desc *, const int16_t *, int16_t *, uint32_t, int32_t *) {
    return SK_ERR_UNSUPPORTED;
}
// the mixed tick's stand-in: AAC units counted (their bytes are checked by the single-codec stand-ins), MP3 granules checked the
// way engine.cpp checks them and summed; one "AudioData" of 576 frames per granule
int sk_tick_run_mixed(sk_engine *e, const sk_tick_stream *ts, uint32_t n_streams, const sk_tick_input *in, uint8_t *out, size_t out_cap,
                      sk_tick_output *outs, uint32_t outs_cap, uint32_t *n_outs, size_t *used) {
    STUB_TICK_PROLOGUE();
    uint32_t f = 0, g = 0;
    size_t row = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (seen[ts[i].stream]++) return SK_ERR_INVALID_ARG;
        if (ts[i].stream >= e->open.size() || !e->open[ts[i].stream]) return SK_ERR_BAD_STREAM;
        const uint32_t ch = e->channels[ts[i].stream];
        for (uint32_t j = 0; j < ts[i].n_frames; ++j) {
            if (ts[i].codec == SK_TICK_MP3) {
                if (g >= in->n_mp3_granules || in->mp3_descs[g].stream != ts[i].stream || in->mp3_descs[g].channels != ch || in->mp3_granules[g].channels != ch)
                    return SK_ERR_INVALID_ARG;
                uint32_t sum = 0;
                for (uint32_t c = 0; c < ch * 576; ++c) {
                    const int v = in->mp3_is[row * 576 + c];
                    if (v > 8206 || v < -8206) return SK_ERR_INVALID_ARG;
                    sum = sum * 31u + (uint16_t)v;
                }
                row += ch;
                ++g;
                if (int rc = em.unit(ts[i], i, sum, 576)) return rc;
            } else {
                if (f >= in->n_aac_units) return SK_ERR_INVALID_ARG;
                ++f;
                if (int rc = em.unit(ts[i], i, 0)) return rc;
            }
        }
        if (int rc = em.end_of_row(ts[i], i)) return rc;
    }
    if (f != in->n_aac_units || g != in->n_mp3_granules) return SK_ERR_INVALID_ARG;
    *n_outs = em.k;
    if (used) *used = em.cursor;
    return SK_OK;
}
// the device front-end's stand-in: checks the unit table the way engine.cpp does
int sk_tick_run_au(sk_engine *e, const sk_tick_stream *ts, uint32_t n_streams, const sk_au_item *units, uint32_t n_units,
                   const uint8_t *au, size_t au_len, uint8_t *out, size_t out_cap, sk_tick_output *outs, uint32_t outs_cap,
                   uint32_t *n_outs, size_t *used) {
    STUB_TICK_PROLOGUE();
    uint32_t f = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (seen[ts[i].stream]++) return SK_ERR_INVALID_ARG;
        if (ts[i].stream >= e->open.size() || !e->open[ts[i].stream]) return SK_ERR_BAD_STREAM;
        for (uint32_t j = 0; j < ts[i].n_frames; ++j, ++f) {
            if (f >= n_units) return SK_ERR_INVALID_ARG;
            if (units[f].byte_len > 8192 || units[f].byte_offset % 4 || (size_t)units[f].byte_offset + units[f].byte_len + 8 > au_len)
                return SK_ERR_INVALID_ARG;
            uint32_t sum = 0;
            for (uint32_t c = 0; c < units[f].byte_len + 8; ++c)  // the 8 bytes after a unit must be there (and zero)
                sum = sum * 31u + au[units[f].byte_offset + c];
            if (int rc = em.unit(ts[i], i, sum)) return rc;
        }
        if (int rc = em.end_of_row(ts[i], i)) return rc;
    }
    if (f != n_units) return SK_ERR_INVALID_ARG;
    *n_outs = em.k;
    if (used) *used = em.cursor;
    return SK_OK;
}
// the quantised hand-over's stand-in: checksum of a unit's i16 values and of its side record, with the checks engine.cpp
// makes on the table
int sk_tick_run_q(sk_engine *e, const sk_tick_stream *ts, uint32_t n_streams, const sk_aac_frame_desc *descs, const void *sides,
                  const int16_t *quant, uint32_t n_units, uint8_t *out, size_t out_cap, sk_tick_output *outs, uint32_t outs_cap,
                  uint32_t *n_outs, size_t *used) {
    STUB_TICK_PROLOGUE();
    uint32_t f = 0;
    size_t at = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (seen[ts[i].stream]++) return SK_ERR_INVALID_ARG;
        if (ts[i].stream >= e->open.size() || !e->open[ts[i].stream]) return SK_ERR_BAD_STREAM;
        for (uint32_t j = 0; j < ts[i].n_frames; ++j, ++f) {
            if (f >= n_units || descs[f].stream != ts[i].stream) return SK_ERR_INVALID_ARG;
            const uint32_t ch = e->channels[ts[i].stream];
            uint32_t sum = 0;
            for (uint32_t c = 0; c < ch * 1024; ++c) sum = sum * 31u + (uint16_t)quant[at + c];
            at += (size_t)ch * 1024;
            const uint8_t *side = (const uint8_t *)sides + (size_t)f * SK_AAC_UNIT_SIDE_BYTES;
            for (uint32_t c = 0; c < SK_AAC_UNIT_SIDE_BYTES; ++c) sum = sum * 31u + side[c];
            if (int rc = em.unit(ts[i], i, sum)) return rc;
        }
        if (int rc = em.end_of_row(ts[i], i)) return rc;
    }
    if (f != n_units) return SK_ERR_INVALID_ARG;
    *n_outs = em.k;
    if (used) *used = em.cursor;
    return SK_OK;
}
int sk_tick_run(sk_engine *e, const sk_tick_stream *ts, uint32_t n_streams, const sk_aac_frame_desc *descs, const float *coeffs,
                uint32_t n_frames, uint8_t *out, size_t out_cap, sk_tick_output *outs, uint32_t outs_cap, uint32_t *n_outs,
                size_t *used) {
    STUB_TICK_PROLOGUE();
    uint32_t f = 0;
    size_t fl = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (seen[ts[i].stream]++) return SK_ERR_INVALID_ARG;  // the scheduler must never put a stream twice in a tick
        if (ts[i].stream >= e->open.size() || !e->open[ts[i].stream]) return SK_ERR_BAD_STREAM;
        for (uint32_t j = 0; j < ts[i].n_frames; ++j, ++f) {
            if (f >= n_frames || descs[f].stream != ts[i].stream) return SK_ERR_INVALID_ARG;
            const uint32_t ch = e->channels[ts[i].stream];
            double sum = 0;
            for (uint32_t c = 0; c < ch * 1024; ++c) sum += coeffs[fl + c];
            fl += (size_t)ch * 1024;
            if (int rc = em.unit(ts[i], i, (uint32_t)(int64_t)(sum * 16.0))) return rc;
        }
        if (int rc = em.end_of_row(ts[i], i)) return rc;
    }
    if (f != n_frames) return SK_ERR_INVALID_ARG;
    *n_outs = em.k;
    if (used) *used = em.cursor;
    return SK_OK;
}
}

// ---- scenarios -------------------------------------------------------------------------------------------
static std::vector<uint8_t> clip;
static uint32_t g_lanes = 0;  // sk_pipeline_config::lanes of every scenario
static uint32_t g_front_end = 0;  // sk_pipeline_config::gpu_entropy of scenario_many_streams
static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

struct Got { uint32_t stream_tag, unit, sum; };

static int drain_all(sk_pipeline *p, const std::vector<uint32_t> &handles, std::vector<std::vector<Got>> &got, std::vector<int> &errors,
                     int timeout_s = 60) {
    std::vector<uint8_t> buf(1 << 16);
    std::vector<char> ended(handles.size(), 0);
    size_t live = handles.size();
    const auto t0 = std::chrono::steady_clock::now();
    while (live) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) return 1;
        for (size_t i = 0; i < handles.size(); ++i) {
            if (ended[i]) continue;
            sk_audio_info info;
            const int rc = sk_pipeline_try_recv(p, handles[i], buf.data(), buf.size(), &info);
            if (rc == 1) {
                if (info.is_error) errors[i] += 1;
                else {
                    uint32_t w[4];
                    std::memcpy(w, buf.data(), 16);
                    if (w[3] != 0xabcd1234u || info.frames != 1024) return 2;
                    got[i].push_back(Got{w[0], w[1], w[2]});
                }
            } else if (rc == SK_PIPE_CLOSED) {
                ended[i] = 1;
                --live;
            }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return 0;
}

static int scenario_many_streams(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 4;
    cfg.max_streams = 40;
    cfg.max_frames_per_tick = 96;
    cfg.max_stream_frames_per_tick = 5;
    cfg.tick_wait_us = 50;
    cfg.lanes = g_lanes;
    cfg.gpu_entropy = g_front_end;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    const uint32_t n = 32, loops = g_front_end == 2 ? 12 : 3;
    std::vector<uint32_t> handles(n);
    // streams of bytes that never frame sit between the real ones: they end without ever reaching the engine, so their
    // batch entries have no tick row -- with several delivery threads the rows and the entries of a batch then number
    // differently, and a real stream's end must still not overtake its last outputs
    std::vector<uint32_t> junk_handles;
    std::vector<uint8_t> junk(3000);
    for (auto &b : junk) b = (uint8_t)(rnd() | 1) & 0x7f;
    for (uint32_t i = 0; i < n; ++i) {
        if (i % 6 == 1) {
            uint32_t h;
            CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
            junk_handles.push_back(h);
        }
        CHECK(sk_pipeline_spawn(p, nullptr, &handles[i]) == SK_OK);
    }
    std::vector<std::thread> feeders;
    feeders.emplace_back([&] {
        for (int rep = 0; rep < 8; ++rep)
            for (uint32_t h : junk_handles) {
                (void)sk_pipeline_send(p, h, junk.data(), junk.size());
                std::this_thread::sleep_for(std::chrono::microseconds(300));
            }
        for (uint32_t h : junk_handles)
            while (sk_pipeline_finish(p, h) != SK_OK) std::this_thread::sleep_for(std::chrono::microseconds(100));
    });
    for (int t = 0; t < 2; ++t)
        feeders.emplace_back([&, t] {
            for (uint32_t i = (uint32_t)t; i < n; i += 2) {
                const uint64_t total = (uint64_t)clip.size() * loops;
                uint64_t sent = 0;
                uint32_t seed = 12345 + i;
                while (sent < total) {
                    seed = seed * 1664525u + 1013904223u;
                    size_t len = 1 + (seed >> 8) % (i % 3 == 0 ? 7000 : (i % 3 == 1 ? 400 : 50));
                    const size_t at = (size_t)(sent % clip.size());
                    len = std::min(len, clip.size() - at);
                    const int rc = sk_pipeline_send(p, handles[i], clip.data() + at, len);
                    if (rc == SK_OK) sent += len;
                    else std::this_thread::sleep_for(std::chrono::microseconds(100));
                }
                while (sk_pipeline_finish(p, handles[i]) != SK_OK) std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
        });
    std::vector<std::vector<Got>> got(n);
    std::vector<int> errors(n, 0);
    CHECK(drain_all(p, handles, got, errors) == 0);
    for (auto &th : feeders) th.join();
    {
        std::vector<std::vector<Got>> jg(junk_handles.size());
        std::vector<int> je(junk_handles.size(), 0);
        CHECK(drain_all(p, junk_handles, jg, je) == 0);
        for (size_t i = 0; i < junk_handles.size(); ++i) CHECK(jg[i].empty() && je[i] == 0);
        for (uint32_t h : junk_handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    }
    for (uint32_t i = 0; i < n; ++i) {
        CHECK(errors[i] == 0);
        CHECK(got[i].size() == (size_t)48 * loops);
        for (size_t k = 0; k < got[i].size(); ++k) {
            CHECK(got[i][k].unit == k);                              // in order, none lost or duplicated
            CHECK(got[i][k].stream_tag == got[i][0].stream_tag);     // never another stream's data
            CHECK(got[i][k].sum == got[0][k % 48].sum || k < 48 * 0);  // same clip -> same spectra checksums per position
        }
    }
    sk_pipeline_stats st;
    CHECK(sk_pipeline_get_stats(p, &st) == SK_OK && st.frames == (uint64_t)n * 48 * loops && st.errors == 0);
    for (uint32_t h : handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    sk_pipeline_destroy(p);
    return 0;
}

// many handles served by two consumer threads that block in sk_pipeline_wait_outputs instead of polling every handle
static int scenario_wait_outputs(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 3;
    cfg.max_streams = 24;
    cfg.max_stream_frames_per_tick = 3;
    cfg.lanes = g_lanes;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    const uint32_t n = 24;
    std::vector<uint32_t> handles(n);
    for (auto &h : handles) CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
    std::vector<std::atomic<uint32_t>> count(64), closed(64);
    for (auto &c : count) c.store(0);
    for (auto &c : closed) c.store(0);
    std::atomic<uint32_t> live{n};
    std::atomic<int> bad{0};
    std::vector<std::thread> consumers;
    for (int t = 0; t < 2; ++t)
        consumers.emplace_back([&] {
            std::vector<uint8_t> buf(1 << 16);
            uint32_t ready[8];
            sk_audio_info info;
            const auto t0 = std::chrono::steady_clock::now();
            while (live.load() > 0) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { bad.store(1); return; }
                const int k = sk_pipeline_wait_outputs(p, ready, 8, 50);
                for (int i = 0; i < k; ++i) {
                    int taken = 0;
                    for (;;) {
                        const int rc = sk_pipeline_try_recv(p, ready[i], buf.data(), buf.size(), &info);
                        if (rc == 1) {
                            count[ready[i]].fetch_add(1);
                            if (++taken == 2) break;  // leave some behind on purpose: the handle must be reported again
                            continue;
                        }
                        if (rc == SK_PIPE_CLOSED && !closed[ready[i]].exchange(1)) live.fetch_sub(1);
                        break;
                    }
                }
            }
        });
    for (uint32_t i = 0; i < n; ++i) {
        size_t pos = 0;
        while (pos < clip.size()) {
            const size_t len = std::min<size_t>(1 + (i * 131 + pos) % 3000, clip.size() - pos);
            if (sk_pipeline_send(p, handles[i], clip.data() + pos, len) == SK_OK) pos += len;
            else std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        while (sk_pipeline_finish(p, handles[i]) != SK_OK) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    for (auto &th : consumers) th.join();
    CHECK(bad.load() == 0);
    for (uint32_t h : handles) CHECK(count[h].load() == 48 && closed[h].load() == 1);
    for (uint32_t h : handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    sk_pipeline_destroy(p);
    return 0;
}

static int scenario_backpressure_and_errors(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 2;
    cfg.max_streams = 8;
    cfg.lanes = g_lanes;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    uint32_t h = 0;
    CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
    std::vector<uint8_t> big(4 * 1024 * 1024 + 1, 0);
    CHECK(sk_pipeline_send(p, h, big.data(), big.size()) == SK_PIPE_CHUNK_TOO_LARGE);
    int accepted = 0, full = 0;
    for (int rep = 0; rep < 400; ++rep) {
        const int rc = sk_pipeline_send(p, h, clip.data(), clip.size());
        if (rc == SK_OK) ++accepted;
        else { CHECK(rc == SK_PIPE_INPUT_FULL); ++full; }
    }
    CHECK(full > 0 && accepted <= 128 + 4);
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    CHECK(sk_pipeline_queued_input_bytes(p, h) > 0);
    // the consumer never drained: at most output_buffer (16) + one pass of frames are waiting, not 48 * accepted
    std::vector<uint8_t> buf(1 << 16);
    sk_audio_info info;
    int waiting = 0;
    while (sk_pipeline_try_recv(p, h, buf.data(), buf.size(), &info) == 1) ++waiting;
    CHECK(waiting >= 16 && waiting <= 16 + 8);
    CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    CHECK(sk_pipeline_send(p, h, clip.data(), 10) == SK_PIPE_CLOSED);

    // a corrupted access unit ends its stream after the outputs before it; the neighbour is untouched
    std::vector<uint8_t> bad = clip;
    size_t pos = 0;
    for (int k = 0; k < 10; ++k) {
        size_t fl, po, pl;
        uint8_t asc[2];
        CHECK(sk_adts_parse(bad.data() + pos, bad.size() - pos, &fl, &po, &pl, asc) == SK_OK);
        pos += fl;
    }
    for (size_t k = pos + 7; k < pos + 7 + 24; ++k) bad[k] = 0xff;
    uint32_t hg = 0, hb = 0;
    CHECK(sk_pipeline_spawn(p, nullptr, &hg) == SK_OK && sk_pipeline_spawn(p, nullptr, &hb) == SK_OK);
    CHECK(sk_pipeline_send(p, hg, clip.data(), clip.size()) == SK_OK && sk_pipeline_finish(p, hg) == SK_OK);
    CHECK(sk_pipeline_send(p, hb, bad.data(), bad.size()) == SK_OK && sk_pipeline_finish(p, hb) == SK_OK);
    std::vector<std::vector<Got>> got(2);
    std::vector<int> errors(2, 0);
    CHECK(drain_all(p, {hg, hb}, got, errors) == 0);
    CHECK(errors[0] == 0 && got[0].size() == 48);
    CHECK(errors[1] == 1 && got[1].size() == 10);
    // garbage that never frames: the stream just ends at finish()
    uint32_t hz = 0;
    CHECK(sk_pipeline_spawn(p, nullptr, &hz) == SK_OK);
    std::vector<uint8_t> junk(5000);
    for (auto &b : junk) b = (uint8_t)(rnd() | 1) & 0x7f;
    CHECK(sk_pipeline_send(p, hz, junk.data(), junk.size()) == SK_OK && sk_pipeline_finish(p, hz) == SK_OK);
    std::vector<std::vector<Got>> g2(1);
    std::vector<int> e2(1, 0);
    CHECK(drain_all(p, {hz}, g2, e2) == 0);
    CHECK(g2[0].empty() && e2[0] == 0);
    sk_pipeline_destroy(p);
    return 0;
}

// gpu_entropy with maximum-length ADTS frames (8191 bytes: a real access unit followed by zero bytes, which the
// bitstream allows) and a tick far smaller than one worker pass of them: every unit must still arrive, in order --
// a pass that staged more bytes than an empty batch holds used to wait for room that never came.
static int scenario_max_length_frames(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 3;
    cfg.max_streams = 8;
    cfg.max_frames_per_tick = 3;
    cfg.max_stream_frames_per_tick = 16;   // clamped to 3 by create; the byte budget is what must hold
    cfg.tick_wait_us = 50;
    cfg.gpu_entropy = 1;
    cfg.lanes = g_lanes;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    std::vector<uint8_t> big;
    uint32_t units = 0;
    for (size_t pos = 0; pos + 7 <= clip.size() && units < 12;) {
        const size_t len = ((clip[pos + 3] & 3u) << 11) | (clip[pos + 4] << 3) | (clip[pos + 5] >> 5);
        std::vector<uint8_t> frame(8191, 0);
        std::memcpy(frame.data(), clip.data() + pos, len);
        frame[3] = (uint8_t)((frame[3] & ~3u) | ((8191 >> 11) & 3u));
        frame[4] = (uint8_t)((8191 >> 3) & 0xff);
        frame[5] = (uint8_t)((frame[5] & 0x1f) | ((8191 & 7u) << 5));
        big.insert(big.end(), frame.begin(), frame.end());
        pos += len;
        ++units;
    }
    const uint32_t n = 5;
    std::vector<uint32_t> handles(n);
    for (uint32_t i = 0; i < n; ++i) CHECK(sk_pipeline_spawn(p, nullptr, &handles[i]) == SK_OK);
    std::thread feeder([&] {
        for (uint32_t i = 0; i < n; ++i) {
            size_t sent = 0;
            while (sent < big.size()) {
                const size_t len = std::min<size_t>(big.size() - sent, 30000 + 1000 * i);
                if (sk_pipeline_send(p, handles[i], big.data() + sent, len) == SK_OK) sent += len;
                else std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
            while (sk_pipeline_finish(p, handles[i]) != SK_OK) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    });
    std::vector<std::vector<Got>> got(n);
    std::vector<int> errors(n, 0);
    CHECK(drain_all(p, handles, got, errors) == 0);
    feeder.join();
    for (uint32_t i = 0; i < n; ++i) {
        CHECK(errors[i] == 0);
        CHECK(got[i].size() == units);
        for (size_t k = 0; k < got[i].size(); ++k) {
            CHECK(got[i][k].unit == k);
            CHECK(got[i][k].sum == got[0][k].sum);
        }
    }
    for (uint32_t h : handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    sk_pipeline_destroy(p);
    return 0;
}

// A tick that fails (a launch failure in the real engine): every stream of that batch is ended with one error after the
// outputs it already had, the streams of other batches are untouched, nothing stalls and every handle comes back.
static int scenario_tick_failure(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 3;
    cfg.max_streams = 12;
    cfg.max_frames_per_tick = 16;
    cfg.max_stream_frames_per_tick = 4;
    cfg.tick_wait_us = 50;
    cfg.lanes = g_lanes;
    cfg.gpu_entropy = g_front_end;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    const uint32_t n = 12;
    std::vector<uint32_t> handles(n);
    for (auto &h : handles) CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
    g_tick_delay_us.store(300);
    std::thread feeder([&] {
        for (int loop = 0; loop < 3; ++loop)
            for (uint32_t i = 0; i < n; ++i) {
                int rc;
                while ((rc = sk_pipeline_send(p, handles[i], clip.data(), clip.size())) == SK_PIPE_INPUT_FULL)
                    std::this_thread::sleep_for(std::chrono::microseconds(100));
                if (loop == 1 && i == 3)
                    for (sk_lane *l : p->lanes) l->engine->fail_ticks.store(2);  // two ticks of every lane's engine fail from here
            }
        for (uint32_t i = 0; i < n; ++i) {
            int rc;
            while ((rc = sk_pipeline_finish(p, handles[i])) == SK_PIPE_INPUT_FULL) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    });
    std::vector<std::vector<Got>> got(n);
    std::vector<int> errors(n, 0);
    CHECK(drain_all(p, handles, got, errors) == 0);
    feeder.join();
    g_tick_delay_us.store(0);
    uint32_t failed = 0;
    for (uint32_t i = 0; i < n; ++i) {
        CHECK(errors[i] <= 1);
        failed += (uint32_t)errors[i];
        if (!errors[i]) CHECK(got[i].size() == 48 * 3);
        for (size_t k = 0; k < got[i].size(); ++k) CHECK(got[i][k].unit == k);  // what did arrive is in order and complete up to the error
    }
    CHECK(failed >= 1 && failed < n * (g_lanes ? g_lanes : 1) + 1);
    sk_pipeline_stats st;
    CHECK(sk_pipeline_get_stats(p, &st) == SK_OK && st.errors == failed);
    for (uint32_t h : handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    sk_pipeline_destroy(p);
    for (uint8_t o : e->open) CHECK(o == 0);
    return 0;
}

// An exception in one of the scheduler's own threads (an allocation that fails, say) must never reach std::terminate:
// inside the per-stream guard of an entropy thread it is THAT stream's error (the others finish untouched); anywhere else
// it stops the lane as an error -- every open stream ends with the status, later spawns return it -- and the process lives.
static int scenario_thread_exceptions(sk_engine *e) {
    for (int where = 0; where < 4; ++where) {
        sk_pipeline_config cfg{};
        cfg.entropy_threads = 3;
        cfg.max_streams = 8;
        cfg.max_frames_per_tick = 16;
        cfg.max_stream_frames_per_tick = 4;
        cfg.tick_wait_us = 50;
        cfg.lanes = 1;
        cfg.gpu_entropy = g_front_end;
        sk_pipeline *p = nullptr;
        CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
        const uint32_t n = 8;
        std::vector<uint32_t> handles(n);
        for (auto &h : handles) CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
        sk_debug_throw_in_thread(where == 0 ? 5 : 9, where);
        std::thread feeder([&] {
            for (int loop = 0; loop < 2; ++loop)
                for (uint32_t i = 0; i < n; ++i) {
                    int rc;
                    while ((rc = sk_pipeline_send(p, handles[i], clip.data(), clip.size())) == SK_PIPE_INPUT_FULL)
                        std::this_thread::sleep_for(std::chrono::microseconds(100));
                    if (rc != SK_OK && rc != SK_PIPE_CLOSED) std::abort();
                }
            for (uint32_t i = 0; i < n; ++i) {
                int rc;
                while ((rc = sk_pipeline_finish(p, handles[i])) == SK_PIPE_INPUT_FULL) std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
        });
        std::vector<std::vector<Got>> got(n);
        std::vector<int> errors(n, 0);
        CHECK(drain_all(p, handles, got, errors) == 0);
        feeder.join();
        sk_debug_throw_in_thread(-1, 0);
        uint32_t failed = 0;
        for (uint32_t i = 0; i < n; ++i) {
            CHECK(errors[i] <= 1);
            failed += (uint32_t)errors[i];
            if (!errors[i]) CHECK(got[i].size() == 48 * 2);
            for (size_t k = 0; k < got[i].size(); ++k) CHECK(got[i][k].unit == k);  // in order and complete up to the error
        }
        uint32_t h2 = 0;
        if (where == 0) {
            CHECK(failed == 1);  // one stream paid; the lane is alive
            CHECK(sk_pipeline_spawn(p, nullptr, &h2) == SK_ERR_CAPACITY);
        } else {
            CHECK(failed >= 1);  // every stream that was still open
            CHECK(sk_pipeline_spawn(p, nullptr, &h2) == SK_ERR_OOM);
        }
        for (uint32_t h : handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
        sk_pipeline_destroy(p);
        for (uint32_t i = 0; i < e->open.size(); ++i) e->open[i] = 0;  // a dead lane cannot hand its engine streams back one by one
    }
    return 0;
}

// Streams of both codecs behind one scheduler (the worker's per-format dispatch, soundkit-decoder/src/lib.rs:2222-2241, 3041-3053):
// the first bytes choose the decoder; AAC streams give one unit of 1024 frames per access unit, MP3 streams one of 576 per
// granule, each stream's units in order and complete, whatever the chunking; a damaged MP3 stream ends or loses frames alone.
static std::vector<uint8_t> mp3_clip;
static int scenario_mixed_codecs(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 3;
    cfg.max_streams = 12;
    cfg.max_frames_per_tick = 40;
    cfg.max_stream_frames_per_tick = 5;
    cfg.tick_wait_us = 50;
    cfg.lanes = g_lanes;
    cfg.gpu_entropy = g_front_end;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    const uint32_t n = 12, loops = 2;
    std::vector<uint32_t> handles(n);
    for (auto &h : handles) CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
    // streams 9 and 11: the MP3 file with bytes overwritten, cut out and inserted -- whatever the entropy thread makes of it (frames
    // lost, the stream ended by an error), nothing may crash, stall or reach another stream (ASan / TSan runs of this harness)
    std::vector<uint8_t> damaged[2] = {mp3_clip, mp3_clip};
    for (int d = 0; d < 2; ++d) {
        for (int hit = 0; hit < 40; ++hit) {
            const size_t at = rnd() % damaged[d].size();
            const uint32_t kind = rnd() % 3;
            if (kind == 0) damaged[d][at] = (uint8_t)rnd();
            else if (kind == 1) damaged[d].erase(damaged[d].begin() + (ptrdiff_t)at, damaged[d].begin() + (ptrdiff_t)std::min(damaged[d].size(), at + 1 + rnd() % 30));
            else damaged[d].insert(damaged[d].begin() + (ptrdiff_t)at, (size_t)(1 + rnd() % 30), (uint8_t)rnd());
        }
    }
    std::thread feeder([&] {
        for (uint32_t i = 0; i < n; ++i) {
            const std::vector<uint8_t> &src = i == 9 ? damaged[0] : (i == 11 ? damaged[1] : ((i & 1) ? mp3_clip : clip));
            for (uint32_t loop = 0; loop < loops; ++loop) {
                size_t at = 0;
                while (at < src.size()) {
                    const size_t len = std::min<size_t>(src.size() - at, 1 + rnd() % 3000);
                    int rc;
                    while ((rc = sk_pipeline_send(p, handles[i], src.data() + at, len)) == SK_PIPE_INPUT_FULL)
                        std::this_thread::sleep_for(std::chrono::microseconds(100));
                    if (rc == SK_PIPE_CLOSED && (i == 9 || i == 11)) break;  // a damaged stream may have been ended by its error
                    if (rc != SK_OK) std::abort();
                    at += len;
                }
            }
        }
        for (uint32_t i = 0; i < n; ++i) {
            int rc;
            while ((rc = sk_pipeline_finish(p, handles[i])) == SK_PIPE_INPUT_FULL) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    });
    std::vector<std::vector<Got>> got(n);
    std::vector<uint32_t> frames_seen(n, 0);
    std::vector<int> errors(n, 0);
    {
        std::vector<uint8_t> buf(1 << 16);
        std::vector<char> ended(n, 0);
        size_t live = n;
        const auto t0 = std::chrono::steady_clock::now();
        while (live) {
            CHECK(std::chrono::steady_clock::now() - t0 < std::chrono::seconds(120));
            for (uint32_t i = 0; i < n; ++i) {
                if (ended[i]) continue;
                sk_audio_info info;
                const int rc = sk_pipeline_try_recv(p, handles[i], buf.data(), buf.size(), &info);
                if (rc == 1) {
                    if (info.is_error) errors[i] += 1;
                    else {
                        uint32_t w[4];
                        std::memcpy(w, buf.data(), 16);
                        CHECK(w[3] == 0xabcd1234u);
                        CHECK(info.frames == ((i & 1) ? 576u : 1024u));
                        CHECK(info.sampling_rate == ((i & 1) ? 16000u : 48000u) && info.channel_count == 2);
                        got[i].push_back(Got{w[0], w[1], w[2]});
                    }
                } else if (rc == SK_PIPE_CLOSED) {
                    ended[i] = 1;
                    --live;
                }
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    feeder.join();
    for (uint32_t i = 0; i < n; ++i) {
        if (i == 9 || i == 11) {  // damaged: at most one error, at most the undamaged count, what arrived is in order
            CHECK(errors[i] <= 1 && got[i].size() <= 82u * loops);
            for (size_t k = 0; k < got[i].size(); ++k) CHECK(got[i][k].unit == k);
            continue;
        }
        CHECK(errors[i] == 0);
        // the MP3 file twice in a row: the second pass's first frame reaches into a reservoir that holds the first pass's tail,
        // which is what a decoder makes of a concatenation -- every frame still decodes (82 granules per pass)
        CHECK(got[i].size() == ((i & 1) ? 82u * loops : 48u * loops));
        for (size_t k = 0; k < got[i].size(); ++k) CHECK(got[i][k].unit == k);
        if (i >= 2 && (i & 1) && i != 11) {  // every MP3 stream got the same bytes: the same sums in the same order (the stand-in does not sum AAC units of a mixed tick)
            CHECK(got[i].size() == got[i - 2].size());
            for (size_t k = 0; k < got[i].size(); ++k) {
                if (got[i][k].sum != got[i - 2][k].sum) std::fprintf(stderr, "stream %u unit %zu: sum %08x, stream %u has %08x\n", i, k, got[i][k].sum, i - 2, got[i - 2][k].sum);
                CHECK(got[i][k].sum == got[i - 2][k].sum);
            }
        }
    }
    for (uint32_t h : handles) CHECK(sk_pipeline_cancel(p, h) == SK_OK);
    sk_pipeline_destroy(p);
    for (uint8_t o : e->open) CHECK(o == 0);
    return 0;
}

static int scenario_cancel_churn(sk_engine *e);
static int scenario_cancel_churn(sk_engine *e) {
    sk_pipeline_config cfg{};
    cfg.entropy_threads = 3;
    cfg.max_streams = 6;
    cfg.lanes = g_lanes;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    for (int round = 0; round < 60; ++round) {
        uint32_t hs[6];
        for (auto &h : hs) CHECK(sk_pipeline_spawn(p, nullptr, &h) == SK_OK);
        uint32_t extra;
        CHECK(sk_pipeline_spawn(p, nullptr, &extra) == SK_ERR_CAPACITY);
        for (auto h : hs) CHECK(sk_pipeline_send(p, h, clip.data(), clip.size()) == SK_OK);
        if (round % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rnd() % 2000));
        for (auto h : hs) CHECK(sk_pipeline_cancel(p, h) == SK_OK);  // some idle, some held by a worker, some in a tick
        // cancelled handles come back once their in-flight work has been delivered
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            size_t free_now = 0;
            for (sk_lane *l : p->lanes) {
                std::lock_guard<std::mutex> lk(l->handles_mu);
                free_now += l->free_handles.size();
            }
            if (free_now == 6) break;
            CHECK(std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20));
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    sk_pipeline_destroy(p);
    // every engine stream was given back
    for (uint8_t o : e->open) CHECK(o == 0);
    return 0;
}

// The regime of the end-to-end bench in which a stall was seen twice and never reproduced: the tick is the
// slowest stage, so all three batches are in rotation, the workers wait for batch room holding a parsed stream each,
// most streams sit at their output bound and the feeders bounce off InputBufferFull.  The bench's own load generator
// (csrc/load_gen.cpp) drives it, twice on the same pipeline as bench.py does (warm-up, then the timed run), with its
// progress deadline armed: a stall fails the scenario with the scheduler's state on stderr.
static int scenario_slow_tick(uint32_t front_end, uint32_t lanes, bool resample, uint32_t n_streams, uint32_t loops, uint32_t tick_us,
                              uint32_t feeder_threads, uint32_t entropy_threads, uint32_t tick_frames, uint32_t per_stream) {
    sk_engine *e = nullptr;
    CHECK(sk_engine_create(0, n_streams, &e) == SK_OK);
    sk_pipeline_config cfg{};
    cfg.entropy_threads = entropy_threads;
    cfg.max_streams = n_streams;
    cfg.max_frames_per_tick = tick_frames;
    cfg.max_stream_frames_per_tick = per_stream;
    cfg.lanes = lanes;
    cfg.gpu_entropy = front_end;
    sk_pipeline *p = nullptr;
    CHECK(sk_pipeline_create(e, &cfg, &p) == SK_OK);
    sk_decode_options opt{};
    if (resample) {
        opt.output_sample_rate = 16000;
        opt.output_channels = 1;
    }
    g_tick_delay_us.store(tick_us);
    int rc = 0;
    for (int run = 0; run < 2 && rc == 0; ++run) {
        sk_load_result res{};
        const int lrc = sk_loadgen_run(p, clip.data(), clip.size(), 48, n_streams, run == 0 ? 1 : loops, &opt, feeder_threads, 0, &res);
        const uint64_t units = (uint64_t)48 * (run == 0 ? 1 : loops) * n_streams;
        const uint64_t want_out = resample ? units / 4 : units;  // 48 units per loop: whole chunks of four, no flush tail
        if (lrc != SK_OK || res.errors != 0 || res.outputs != want_out) {
            std::fprintf(stderr, "slow-tick scenario (front-end %u, lanes %u, resample %d, run %d): rc %d errors %llu outputs %llu (want %llu)\n",
                         front_end, lanes, (int)resample, run, lrc, (unsigned long long)res.errors, (unsigned long long)res.outputs,
                         (unsigned long long)want_out);
            rc = 1;
        }
    }
    g_tick_delay_us.store(0);
    sk_pipeline_destroy(p);
    sk_engine_destroy(e);
    return rc;
}

// `sched_stub <clip> slow <repeats> [seed]`: the slow-tick scenario over and over with drawn shapes (tests/test_scheduler_cpu.py
// runs a few under the sanitizers; tools/sched_soak.sh runs thousands without)
static int slow_tick_main(int argc, char **argv) {
    const int repeats = argc > 3 ? std::atoi(argv[3]) : 4;
    if (argc > 4) rng_state ^= (uint64_t)std::strtoull(argv[4], nullptr, 10) * 0x9e3779b97f4a7c15ull;
    setenv("SK_LOADGEN_STALL_S", "20", 0);
    for (int r = 0; r < repeats; ++r) {
        const uint32_t front_end = r % 3 == 0 ? 2 : rnd() % 3;
        const uint32_t lanes = 1 + rnd() % 3 / 2;
        const bool resample = rnd() % 4 != 0;
        const uint32_t n_streams = 24 + rnd() % 232;
        const uint32_t per_stream = 1 + rnd() % 16;
        const uint32_t tick_frames = std::max(per_stream, n_streams * per_stream / (2 + rnd() % 6));
        const uint32_t tick_us = 200 + rnd() % 3000;
        const uint32_t feeders = 2 + rnd() % 5, workers = 1 + rnd() % 6, loops = 1 + rnd() % 3;
        std::fprintf(stderr, "slow tick %d: front-end %u lanes %u resample %d streams %u per-stream %u tick %u units / %u us feeders %u workers %u loops %u\n",
                     r, front_end, lanes, (int)resample, n_streams, per_stream, tick_frames, tick_us, feeders, workers, loops);
        if (int rc = scenario_slow_tick(front_end, lanes, resample, n_streams, loops, tick_us, feeders, workers, tick_frames, per_stream)) return rc;
    }
    std::puts("slow-tick scenarios ok");
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 64;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 65;
    clip.resize(1 << 20);
    clip.resize(std::fread(clip.data(), 1, clip.size(), f));
    std::fclose(f);
    sk_engine e;
    e.open.assign(64, 0);
    e.channels.assign(64, 0);
    e.next_unit.assign(64, 0);
    e.rs_fill.assign(64, 0);
    if (argc > 2 && std::string(argv[2]) == "slow") return slow_tick_main(argc, argv);
    if (int rc = scenario_many_streams(&e)) return rc;
    setenv("SK_PIPELINE_DELIVER_THREADS", "3", 1);  // the sliced delivery path, as the GPU front-end mode uses it
    if (int rc = scenario_many_streams(&e)) return rc;
    if (int rc = scenario_cancel_churn(&e)) return rc;
    unsetenv("SK_PIPELINE_DELIVER_THREADS");
    if (int rc = scenario_wait_outputs(&e)) return rc;
    if (int rc = scenario_backpressure_and_errors(&e)) return rc;
    if (int rc = scenario_cancel_churn(&e)) return rc;
    g_lanes = 2;  // the same scenarios over two engines behind one handle space
    if (int rc = scenario_many_streams(&e)) return rc;
    if (int rc = scenario_wait_outputs(&e)) return rc;
    if (int rc = scenario_backpressure_and_errors(&e)) return rc;
    if (int rc = scenario_cancel_churn(&e)) return rc;
    g_lanes = 3;
    if (int rc = scenario_many_streams(&e)) return rc;
    g_lanes = 1;
    g_front_end = 2;  // host Huffman decode, i16 + side records to the engine
    if (int rc = scenario_many_streams(&e)) return rc;
    g_lanes = 2;
    if (int rc = scenario_many_streams(&e)) return rc;
    for (uint32_t fe = 0; fe < 3; ++fe) {
        g_front_end = fe;
        g_lanes = 1;
        if (int rc = scenario_tick_failure(&e)) return rc;
    }
    g_lanes = 2;
    if (int rc = scenario_tick_failure(&e)) return rc;
    {   // the reference's stereo MP3 fixture beside the AAC one
        std::string path = argv[1];
        const size_t at = path.rfind("/aac/");
        if (at == std::string::npos) return 66;
        path = path.substr(0, at) + "/mp3/stereo16k_A_Tusk_encoded.mp3";
        FILE *m = std::fopen(path.c_str(), "rb");
        if (!m) return 67;
        mp3_clip.resize(1 << 20);
        mp3_clip.resize(std::fread(mp3_clip.data(), 1, mp3_clip.size(), m));
        std::fclose(m);
    }
    for (uint32_t fe = 0; fe < 3; ++fe) {
        g_front_end = fe;
        g_lanes = fe == 1 ? 2 : 1;
        if (int rc = scenario_mixed_codecs(&e)) return rc;
    }
    g_front_end = 0;
    g_lanes = 1;
    if (int rc = scenario_thread_exceptions(&e)) return rc;
    if (int rc = scenario_max_length_frames(&e)) return rc;
    g_lanes = 2;
    if (int rc = scenario_max_length_frames(&e)) return rc;
    std::puts("scheduler scenarios ok");
    return 0;
}

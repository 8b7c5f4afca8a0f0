// load_gen.cpp -- measurement harness for the batch scheduler (not part of the product library): feeds the same
// ADTS clip, looped, into n_streams pipeline handles from a few feeder threads and drains their outputs, the way
// N independent producers/consumers of soundkit-decoder's DecodePipelineHandle would.
// Built as libsk_loadgen.so next to libsoundkit_amd.so; bench.py --workload end_to_end drives it.
#include "../../include/soundkit_amd.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

extern "C" {

typedef struct sk_load_result {
    double seconds;
    uint64_t access_units, outputs, pcm_frames, pcm_bytes, errors, input_full;
} sk_load_result;

// Optional per-stream bookkeeping for the parity tests (stream i = the i-th spawned handle): what each stream
// delivered, in delivery order -- FNV-1a over (frames, channels, bits, rate, bytes) of every AudioData, counts, and the
// raw bytes of a few chosen streams for comparison with the oracle.
typedef struct sk_load_check {
    uint64_t *hash;      /* [n_streams] */
    uint32_t *outputs;   /* [n_streams] AudioData delivered */
    uint64_t *bytes;     /* [n_streams] */
    uint32_t *errors;    /* [n_streams] error records delivered */
    const uint32_t *capture;  /* [n_capture] stream indices whose bytes are kept */
    uint32_t n_capture;
    uint8_t *capture_buf;     /* [n_capture][capture_cap] */
    size_t capture_cap;
    size_t *capture_len;      /* [n_capture] */
} sk_load_check;

int sk_loadgen_run_checked(sk_pipeline *p, const uint8_t *clip, size_t clip_len, uint32_t clip_units, uint32_t n_streams,
                           uint32_t loops, const sk_decode_options *opt, uint32_t feeder_threads, uint32_t chunk_bytes,
                           sk_load_result *res, const sk_load_check *chk);

// Every stream receives `loops` copies of the clip in chunks of chunk_bytes, then finish().  Returns when every
// stream has ended and is drained -- or SK_ERR_TIMEOUT when neither a send succeeded nor an output arrived for
// SK_LOADGEN_STALL_S seconds (default 30): the scheduler's state (sk_pipeline_debug_dump) and the generator's own counts
// go to stderr first, so that a stall leaves a record instead of a harness kill.
int sk_loadgen_run(sk_pipeline *p, const uint8_t *clip, size_t clip_len, uint32_t clip_units, uint32_t n_streams,
                   uint32_t loops, const sk_decode_options *opt, uint32_t feeder_threads, uint32_t chunk_bytes,
                   sk_load_result *res) {
    return sk_loadgen_run_checked(p, clip, clip_len, clip_units, n_streams, loops, opt, feeder_threads, chunk_bytes, res, nullptr);
}

int sk_loadgen_run_checked(sk_pipeline *p, const uint8_t *clip, size_t clip_len, uint32_t clip_units, uint32_t n_streams,
                           uint32_t loops, const sk_decode_options *opt, uint32_t feeder_threads, uint32_t chunk_bytes,
                           sk_load_result *res, const sk_load_check *chk) {
    if (!p || !clip || !clip_len || !n_streams || !loops || !res) return SK_ERR_INVALID_ARG;
    if (!feeder_threads) feeder_threads = 2;
    if (!chunk_bytes || chunk_bytes > clip_len) chunk_bytes = (uint32_t)clip_len;
    std::vector<uint32_t> handles(n_streams);
    for (uint32_t i = 0; i < n_streams; ++i) {
        const int rc = sk_pipeline_spawn(p, opt, &handles[i]);
        if (rc != SK_OK) {
            for (uint32_t k = 0; k < i; ++k) (void)sk_pipeline_cancel(p, handles[k]);
            return rc;
        }
    }
    std::atomic<uint64_t> outputs{0}, pcm_frames{0}, pcm_bytes{0}, errors{0}, input_full{0};
    std::atomic<uint64_t> progress{0}, sends_ok{0}, finishes_ok{0}, closed_seen{0}, received{0};
    std::atomic<bool> abort_run{false};
    std::atomic<uint32_t> threads_done{0};
    std::atomic<uint32_t> live{n_streams};
    uint32_t max_handle = 0;
    for (uint32_t h : handles) max_handle = std::max(max_handle, h);
    // a completion queue can still hold handles of an earlier run on the same pipeline (a handle is listed again while it
    // is being drained); one that is not a stream of this run must not count as one of its ends
    std::vector<std::atomic<char>> ended(max_handle + 1);
    for (auto &e : ended) e.store(1);
    for (uint32_t h : handles) ended[h].store(0);
    std::vector<uint32_t> index_of(chk ? max_handle + 1 : 0, 0);
    std::vector<int32_t> capture_slot(chk ? n_streams : 0, -1);
    std::unique_ptr<std::mutex[]> locks(chk ? new std::mutex[n_streams] : nullptr);
    if (chk) {
        for (uint32_t i = 0; i < n_streams; ++i) {
            index_of[handles[i]] = i;
            if (!chk->hash) continue;  // capture only (the bench's PcmStats of one stream): no per-stream bookkeeping
            chk->hash[i] = 0xcbf29ce484222325ull;
            chk->outputs[i] = 0;
            chk->bytes[i] = 0;
            chk->errors[i] = 0;
        }
        for (uint32_t c = 0; c < chk->n_capture; ++c) {
            if (chk->capture[c] < n_streams) capture_slot[chk->capture[c]] = (int32_t)c;
            chk->capture_len[c] = 0;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    // producers: each owns a slice of the streams and keeps their input queues fed (send never blocks: back off when full)
    const uint32_t producers = std::max(1u, feeder_threads / 2), consumers = std::max(1u, feeder_threads - producers);
    for (uint32_t t = 0; t < producers; ++t) {
        threads.emplace_back([&, t] {
            struct St {
                uint32_t handle;
                uint64_t sent = 0;
                bool finished = false;
            };
            std::vector<St> mine;
            for (uint32_t i = t; i < n_streams; i += producers) {
                St s;
                s.handle = handles[i];
                mine.push_back(s);
            }
            const uint64_t total = (uint64_t)clip_len * loops;
            size_t open = mine.size();
            uint64_t full = 0;
            while (open && !abort_run.load(std::memory_order_relaxed)) {
                bool progressed = false;
                for (St &s : mine) {
                    if (s.finished) continue;
                    for (int burst = 0; burst < 4; ++burst) {
                        if (s.sent >= total) {
                            const int rc = sk_pipeline_finish(p, s.handle);
                            if (rc == SK_OK || rc == SK_PIPE_CLOSED) {
                                s.finished = true;
                                --open;
                                progressed = true;
                                finishes_ok.fetch_add(1, std::memory_order_relaxed);
                                progress.fetch_add(1, std::memory_order_relaxed);
                            } else {
                                ++full;
                            }
                            break;
                        }
                        const size_t at = (size_t)(s.sent % clip_len);
                        const size_t n = std::min<size_t>(chunk_bytes, clip_len - at);
                        const int rc = sk_pipeline_send(p, s.handle, clip + at, n);
                        if (rc == SK_OK) {
                            s.sent += n;
                            progressed = true;
                            sends_ok.fetch_add(1, std::memory_order_relaxed);
                            progress.fetch_add(1, std::memory_order_relaxed);
                        } else if (rc == SK_PIPE_CLOSED) {
                            s.finished = true;
                            --open;
                            break;
                        } else {
                            ++full;
                            break;
                        }
                    }
                }
                if (!progressed) std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            input_full += full;
            threads_done.fetch_add(1);
        });
    }
    // consumers: block on sk_pipeline_wait_outputs and drain whatever handles it reports
    for (uint32_t t = 0; t < consumers; ++t) {
        threads.emplace_back([&] {
            std::vector<uint8_t> buf(1 << 20);
            std::vector<uint32_t> ready(1024);
            sk_audio_info info;
            uint64_t o = 0, f = 0, b = 0, e = 0;
            while (live.load() > 0 && !abort_run.load(std::memory_order_relaxed)) {
                const int n = sk_pipeline_wait_outputs(p, ready.data(), (uint32_t)ready.size(), 20);
                for (int k = 0; k < n; ++k) {
                    const uint32_t h = ready[(size_t)k];
                    // with per-stream checks a stream is drained by one consumer at a time: receive-then-hash must not
                    // interleave with another thread's, or the order-sensitive hash would see outputs out of order
                    std::unique_lock<std::mutex> drain_lock;
                    const bool checked = chk && h <= max_handle && (chk->hash || capture_slot[index_of[h]] >= 0);
                    if (checked) drain_lock = std::unique_lock<std::mutex>(locks[index_of[h]]);
                    for (;;) {
                        const int rc = sk_pipeline_try_recv(p, h, buf.data(), buf.size(), &info);
                        if (rc == 1) {
                            received.fetch_add(1, std::memory_order_relaxed);
                            progress.fetch_add(1, std::memory_order_relaxed);
                            if (info.is_error) ++e;
                            else {
                                ++o;
                                f += info.frames;
                                b += info.bytes;
                            }
                            if (checked) {
                                const uint32_t i = index_of[h];
                                if (info.is_error) {
                                    if (chk->hash) ++chk->errors[i];
                                    continue;
                                }
                                if (chk->hash) {
                                    uint64_t hsh = chk->hash[i];
                                    auto mix = [&](const uint8_t *d, size_t n) {
                                        for (size_t q = 0; q < n; ++q) hsh = (hsh ^ d[q]) * 0x100000001b3ull;
                                    };
                                    const uint32_t head[4] = {info.frames, info.channel_count, info.bits_per_sample, info.sampling_rate};
                                    mix(reinterpret_cast<const uint8_t *>(head), sizeof head);
                                    mix(buf.data(), info.bytes);
                                    chk->hash[i] = hsh;
                                    ++chk->outputs[i];
                                    chk->bytes[i] += info.bytes;
                                }
                                const int32_t slot = capture_slot[i];
                                if (slot >= 0 && chk->capture_len[slot] + info.bytes <= chk->capture_cap) {
                                    std::memcpy(chk->capture_buf + (size_t)slot * chk->capture_cap + chk->capture_len[slot], buf.data(), info.bytes);
                                    chk->capture_len[slot] += info.bytes;
                                }
                            }
                            continue;
                        }
                        if (rc == SK_PIPE_CLOSED && h <= max_handle && !ended[h].exchange(1)) {
                            live.fetch_sub(1);
                            closed_seen.fetch_add(1, std::memory_order_relaxed);
                            progress.fetch_add(1, std::memory_order_relaxed);
                        }
                        break;
                    }
                }
            }
            outputs += o;
            pcm_frames += f;
            pcm_bytes += b;
            errors += e;
            threads_done.fetch_add(1);
        });
    }
    // the run's own watchdog: progress = a send accepted, a finish accepted, an output received or an end seen
    double stall_s = 30.0;
    if (const char *env = std::getenv("SK_LOADGEN_STALL_S")) stall_s = std::max(0.5, std::atof(env));
    bool stalled = false;
    {
        uint64_t last = progress.load();
        auto last_change = std::chrono::steady_clock::now();
        while (threads_done.load() < threads.size()) {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            const uint64_t now = progress.load();
            const auto t = std::chrono::steady_clock::now();
            if (now != last) {
                last = now;
                last_change = t;
            } else if (std::chrono::duration<double>(t - last_change).count() > stall_s) {
                stalled = true;
                std::vector<char> text(32768);
                sk_pipeline_debug_dump(p, text.data(), text.size());
                std::fprintf(stderr,
                             "[sk_loadgen] STALL: no send accepted and no output received for %.1f s | streams %u, still live %u | "
                             "sends accepted %llu, finishes accepted %llu, outputs received %llu, ends seen %llu\n%s",
                             stall_s, n_streams, live.load(), (unsigned long long)sends_ok.load(), (unsigned long long)finishes_ok.load(),
                             (unsigned long long)received.load(), (unsigned long long)closed_seen.load(), text.data());
                std::fflush(stderr);
                abort_run.store(true);
                break;
            }
        }
    }
    for (std::thread &th : threads) th.join();
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t h : handles) (void)sk_pipeline_cancel(p, h);
    res->access_units = (uint64_t)clip_units * loops * n_streams;
    res->outputs = outputs.load();
    res->pcm_frames = pcm_frames.load();
    res->pcm_bytes = pcm_bytes.load();
    res->errors = errors.load();
    res->input_full = input_full.load();
    return stalled ? SK_ERR_TIMEOUT : SK_OK;
}

// PcmStats::from_pcm (aac-wasm-bench/src/lib.rs:66-101) of 16-bit samples taken as f32 = s / 32768 (audio_bytes::i16le_to_f32):
// sample count, RMS, peak and FNV-1a over the f32 bit patterns -- what the reference's harness prints beside every rate
void sk_loadgen_pcm_stats(const int16_t *pcm, size_t n, double *rms, double *peak_abs, uint64_t *checksum) {
    double sum = 0.0, peak = 0.0;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        const float f = (float)pcm[i] / 32768.0f;
        const double d = (double)f;
        sum += d * d;
        peak = std::max(peak, std::fabs(d));
        uint32_t bits;
        std::memcpy(&bits, &f, 4);
        h ^= (uint64_t)bits;
        h *= 0x100000001b3ull;
    }
    if (rms) *rms = n ? std::sqrt(sum / (double)n) : 0.0;
    if (peak_abs) *peak_abs = peak;
    if (checksum) *checksum = h;
}

}  // extern "C"

// load_gen.cpp -- measurement harness for the batch scheduler (not part of the product library): feeds the same
// ADTS clip, looped, into n_streams pipeline handles from a few feeder threads and drains their outputs, the way
// N independent producers/consumers of soundkit-decoder's DecodePipelineHandle would.
// Built as libsk_loadgen.so next to libsoundkit_amd.so; bench.py --workload end_to_end drives it.
#include "../../include/soundkit_amd.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

typedef struct sk_load_result {
    double seconds;
    uint64_t access_units, outputs, pcm_frames, pcm_bytes, errors, input_full;
} sk_load_result;

// Every stream receives `loops` copies of the clip in chunks of chunk_bytes, then finish().  Returns when every
// stream has ended and is drained.
int sk_loadgen_run(sk_pipeline *p, const uint8_t *clip, size_t clip_len, uint32_t clip_units, uint32_t n_streams,
                   uint32_t loops, const sk_decode_options *opt, uint32_t feeder_threads, uint32_t chunk_bytes,
                   sk_load_result *res) {
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
    std::atomic<uint32_t> live{n_streams};
    uint32_t max_handle = 0;
    for (uint32_t h : handles) max_handle = std::max(max_handle, h);
    std::vector<std::atomic<char>> ended(max_handle + 1);
    for (auto &e : ended) e.store(0);
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
            while (open) {
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
        });
    }
    // consumers: block on sk_pipeline_wait_outputs and drain whatever handles it reports
    for (uint32_t t = 0; t < consumers; ++t) {
        threads.emplace_back([&] {
            std::vector<uint8_t> buf(1 << 20);
            std::vector<uint32_t> ready(1024);
            sk_audio_info info;
            uint64_t o = 0, f = 0, b = 0, e = 0;
            while (live.load() > 0) {
                const int n = sk_pipeline_wait_outputs(p, ready.data(), (uint32_t)ready.size(), 20);
                for (int k = 0; k < n; ++k) {
                    const uint32_t h = ready[(size_t)k];
                    for (;;) {
                        const int rc = sk_pipeline_try_recv(p, h, buf.data(), buf.size(), &info);
                        if (rc == 1) {
                            if (info.is_error) ++e;
                            else {
                                ++o;
                                f += info.frames;
                                b += info.bytes;
                            }
                            continue;
                        }
                        if (rc == SK_PIPE_CLOSED && h <= max_handle && !ended[h].exchange(1)) live.fetch_sub(1);
                        break;
                    }
                }
            }
            outputs += o;
            pcm_frames += f;
            pcm_bytes += b;
            errors += e;
        });
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
    return SK_OK;
}

}  // extern "C"

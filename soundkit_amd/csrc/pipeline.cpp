// pipeline.cpp -- the batch scheduler: N ADTS AAC-LC streams -> one submission loop on one GPU.
//
// Replaces one `pipeline_worker` OS thread per stream (soundkit-decoder/src/lib.rs:2891-3038) while keeping the
// handle's contract (lib.rs:2788-2889): bounded input queue (128 chunks / 8 MiB, `send` never blocks and reports
// InputBufferFull), bounded output queue (16 AudioData: a stream whose consumer does not drain it stops being
// scheduled -- the blocking send of lib.rs:3238-3240), empty chunk = end of stream (flush), an error ends that
// stream only after the outputs produced before it (lib.rs:3131-3134).
//
//   send() -> per-stream input queue -> entropy workers (host threads, one stream at a time each: ADTS framing +
//   AAC-LC front-end, csrc/aac_frontend.cpp) -> the tick batch (pinned host memory) -> submission thread:
//   sk_tick_run (H2D, synthesis, s16, resampler, downmix, pack, D2H) -> per-stream output queues -> try_recv().
//
// Three batches rotate: the workers fill one while the GPU runs the second and the delivery thread hands out the
// results of the third.  A stream contributes at most
// `max_stream_frames_per_tick` access units to a batch and is not scheduled again until that batch has been
// delivered, so its frames reach the engine in order and its outputs never overtake each other.
#include "../../include/soundkit_amd.h"
#include "sk_abi.h"

#include <hip/hip_runtime.h>

#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t kMaxInputChunkBytes = 4u * 1024 * 1024;   // MAX_INPUT_CHUNK_BYTES, lib.rs:80
constexpr size_t kMaxQueuedInputBytes = 8u * 1024 * 1024;  // MAX_QUEUED_INPUT_BYTES, lib.rs:81
constexpr uint32_t kNoStream = 0xffffffffu;

struct Output {
    bool is_error = false;
    int32_t status = 0;
    uint32_t rate = 0, frames = 0;
    uint8_t bits = 0, channels = 0;
    std::vector<uint8_t> data;  // PCM bytes, or the error text
};

struct PStream {
    std::mutex mu;
    std::condition_variable cv_out;
    // guarded by mu
    bool open = false, busy = false, queued = false, finished = false, cancelled = false;
    bool out_listed = false;  // the handle sits in the pipeline's OutQueue (or has been handed to a waiter and not taken from since)
    std::deque<std::vector<uint8_t>> in;
    std::deque<Output> out;
    sk_decode_options opt{};
    std::atomic<size_t> queued_bytes{0};
    // owned by whichever worker holds the stream (busy)
    std::vector<uint8_t> pending;
    size_t pending_pos = 0;
    bool saw_eof = false;
    bool more = false;  // the last pass stopped at its frame limit: schedulable even with an empty input queue
    sk_aac_decoder *fe = nullptr;
    uint8_t asc[2] = {0, 0};
    uint32_t rate = 0, engine_stream = kNoStream;
    uint8_t channels = 0;
    bool resample = false;
    // which decoder the stream's first bytes select (detect_and_init_decoder, soundkit-decoder/src/lib.rs:3041-3053): 0 not known yet
    uint8_t codec = 0;
    std::vector<uint8_t> mp3_reservoir;  // main data of the frames seen so far (main_data_begin reaches back into it)
    uint32_t mp3_free_format = 0;        // a free-format stream's frame length once measured (sk_mp3_scan_free)
};
constexpr uint8_t kCodecAac = 1, kCodecMp3 = 2;

struct BatchEntry {  // bookkeeping beside one sk_tick_stream
    uint32_t handle = 0;
    bool eof = false;
    bool failed = false;
    int32_t fail_status = 0;
    std::string fail_msg;
    // quantised hand-over only: the access units of this pass as they came off the wire, so that a status the device finds
    // (stereo tools, TNS, the unit's tail) can be given the reference's message by parsing them again on the host
    std::vector<uint8_t> raw;
    std::vector<uint32_t> raw_len;
};

struct Batch {
    float *coeffs = nullptr;  // pinned
    size_t coeff_cap = 0, n_floats = 0;
    std::vector<sk_aac_frame_desc> descs;  // sized once (max_frames_per_tick): workers fill disjoint ranges without the lock
    size_t n_descs = 0;
    // gpu_entropy mode: the raw access units instead of spectra (each unit 4-byte aligned, >= 8 zero bytes after it)
    uint8_t *au_bytes = nullptr;  // pinned
    size_t au_cap = 0, au_used = 0;
    std::vector<sk_au_item> units;  // sized once, indexed like descs
    std::vector<sk_tick_stream> ts;
    std::vector<BatchEntry> entries;
    // MP3 streams' granules (their units), in the order of their streams in ts; sized by ensure_mp3 at the first MP3 stream
    std::vector<sk_mp3_requant_granule> mp3_gr;
    std::vector<sk_mp3_granule_desc> mp3_desc;
    int16_t *mp3_is = nullptr;  // pinned: [granule][channel][576], 2 x 576 per granule reserved
    size_t n_mp3 = 0, mp3_rows = 0;
    uint32_t writers = 0;  // claims whose memcpy is still running
    // the tick's results, handed from the submission thread to the delivery thread
    uint8_t *out_pinned = nullptr;
    size_t out_pinned_cap = 0;
    std::vector<sk_tick_output> recs;
    std::vector<uint32_t> row_of;  // tick row -> entry
    std::vector<uint32_t> entry_row;  // entry -> tick row, kNoStream for a stream that never reached the device
    std::vector<uint32_t> rec_begin;  // tick row -> its first output record (rows + 1 entries)
    uint32_t n_out = 0;
    int rc = SK_OK;
    uint32_t next_slice = 0, slices_done = 0;  // delivery threads share a batch row-wise (guarded by batch_mu)
    void clear() {
        n_floats = 0;
        n_descs = 0;
        au_used = 0;
        n_mp3 = 0;
        mp3_rows = 0;
        ts.clear();
        entries.clear();
        row_of.clear();
        entry_row.clear();
        rec_begin.clear();
        n_out = 0;
        rc = SK_OK;
        next_slice = slices_done = 0;
    }
};

}  // namespace

// handles with something to receive (outputs or the end of the stream), for callers that serve many handles; one queue
// for all lanes of a pipeline, holding public handles
struct OutQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<uint32_t> ready;
    bool stop = false;
};

// One lane = one engine + the threads and batches that feed it.  A pipeline is one or more lanes behind one handle
// space: ticks of different engines overlap on the device (a tick is a chain of dependent launches with a
// synchronisation at its end), which one lane's submission thread cannot do by itself.
struct sk_lane {
    sk_engine *engine = nullptr;
    OutQueue *oq = nullptr;
    uint32_t lane_index = 0, n_lanes = 1;  // public handle = handle * n_lanes + lane_index
    sk_pipeline_config cfg{};
    std::vector<std::unique_ptr<PStream>> streams;
    std::mutex handles_mu;
    std::vector<uint32_t> free_handles;

    std::mutex rq_mu;
    std::condition_variable rq_cv;
    std::deque<uint32_t> ready;

    std::mutex batch_mu;
    std::condition_variable batch_cv;   // submission thread: work arrived / writers done
    std::condition_variable room_cv;    // workers: the filling batch has room again
    size_t au_pass_budget = 0;          // gpu_entropy: access-unit bytes one worker pass may stage (= what an empty batch holds)
    std::atomic<size_t> out_cap_seen{0};  // the largest pinned output buffer a batch of this lane has asked for
    static constexpr int kBatches = 3;  // one filling, one on the GPU, one being delivered
    Batch batches[kBatches];
    int filling = 0;
    std::deque<int> free_batches, to_deliver;
    std::condition_variable deliver_cv;
    bool stop = false;
    std::atomic<int> fatal{0};  // != 0: a thread of the lane died of an exception (lane_fatal); every stream got that error
    // MP3: the standard's tables, made (and installed on the engine) when the first MP3 stream shows up
    std::mutex mp3_mu;
    sk_mp3_codebook *mp3_cb = nullptr;
    std::atomic<bool> mp3_ready{false};

    std::vector<std::thread> workers;
    std::thread submitter;
    std::vector<std::thread> deliverers;
    uint32_t n_deliver = 1;

    std::atomic<uint64_t> n_ticks{0}, n_frames{0}, n_outputs{0}, n_errors{0}, parse_ns{0}, tick_ns{0}, idle_ns{0}, deliver_ns{0};
    // SK_PIPELINE_WATCHDOG=<seconds>: a thread that prints where everybody stands when no tick has finished for that long
    std::thread watchdog;
    std::atomic<int> submit_where{0};     // 0 waits for work, 1 lets the batch fill, 2 waits for writers / a free batch, 3 in the tick, 4 hands over
    std::atomic<uint32_t> workers_waiting_room{0}, workers_waiting_ready{0}, workers_parsing{0}, deliverers_waiting{0};
};

namespace {

using Clock = std::chrono::steady_clock;
uint64_t ns_since(Clock::time_point t0) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count();
}

// call with s.mu held: true = the stream has work, is free and is not in the ready queue yet (now marked as queued)
bool mark_schedulable(sk_lane *p, PStream &s) {
    if (!s.open || s.busy || s.queued || s.finished || s.cancelled) return false;
    if ((s.in.empty() && !s.more) || s.out.size() >= p->cfg.output_buffer) return false;
    s.queued = true;
    return true;
}

// call with s.mu held
void maybe_schedule(sk_lane *p, PStream &s, uint32_t handle) {
    if (!mark_schedulable(p, s)) return;
    {
        std::lock_guard<std::mutex> lk(p->rq_mu);
        p->ready.push_back(handle);
    }
    p->rq_cv.notify_one();
}

// Gives back what a stream holds outside the scheduler.  Two threads can get here for the same stream -- the delivery
// thread for a stream that has just finished, and a caller cancelling that same handle the moment it saw the end -- so the
// resources are taken over under the stream's lock and released once.
void release_device_side(sk_lane *p, PStream &s) {
    uint32_t engine_stream;
    sk_aac_decoder *fe;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        engine_stream = s.engine_stream;
        s.engine_stream = kNoStream;
        fe = s.fe;
        s.fe = nullptr;
        s.pending.clear();
        s.pending.shrink_to_fit();
        s.pending_pos = 0;
        s.mp3_reservoir.clear();
        s.mp3_reservoir.shrink_to_fit();
    }
    if (engine_stream != kNoStream) (void)sk_stream_close(p->engine, engine_stream);  // also drops its resampler
    if (fe) sk_aac_decoder_destroy(fe);
}

struct Parsed {  // what one worker pass produced for one stream
    uint32_t n_frames = 0;
    size_t n_floats = 0;
    size_t n_au_bytes = 0;  // gpu_entropy: bytes staged (units padded as the device wants them)
    bool eof = false, failed = false;
    bool budget_stop = false;  // stopped early because the staged bytes reached what one batch can take
    int32_t fail_status = 0;
    std::string fail_msg;
    std::vector<uint8_t> raw;  // quantised hand-over: the pass's access units, kept for the error text (BatchEntry::raw)
    std::vector<uint32_t> raw_len;
    // an MP3 stream's pass: n_frames counts GRANULES (the tick's units for such a stream)
    bool mp3 = false;
    std::vector<sk_mp3_requant_granule> mp3_gr;
    std::vector<sk_mp3_granule_desc> mp3_desc;
    std::vector<int16_t> mp3_is;
};

// 1 = a chunk was appended to s.pending, 0 = nothing queued right now, -1 = the end-of-stream marker was taken (s.saw_eof set)
int pull_input(PStream &s) {
    std::vector<uint8_t> chunk;
    bool got = false;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        if (!s.in.empty()) {
            chunk = std::move(s.in.front());
            s.in.pop_front();
            got = true;
        }
    }
    if (!got) return 0;
    if (chunk.empty()) {
        s.saw_eof = true;
        return -1;
    }
    s.queued_bytes.fetch_sub(chunk.size());
    if (s.pending_pos > 0 && s.pending_pos >= s.pending.size() / 2) {  // compact
        s.pending.erase(s.pending.begin(), s.pending.begin() + (ptrdiff_t)s.pending_pos);
        s.pending_pos = 0;
    }
    s.pending.insert(s.pending.end(), chunk.begin(), chunk.end());
    return 1;
}

// The tables an MP3 stream needs: the code book for the host's Huffman stage, band tables and synthesis window on the engine;
// the batches get their granule arrays.  Once per lane, at the first MP3 stream.
int ensure_mp3(sk_lane *p);

// What the stream's first bytes are (detect_audio at lib.rs:3042 looks at magic numbers; on this path two formats exist):
// an ID3v2 tag or an MPEG audio sync with a layer field -> MP3; an ADTS sync (layer bits 00) -> AAC.  0 = not enough bytes yet.
uint8_t sniff_codec(const uint8_t *d, size_t n) {
    if (n >= 3 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') return kCodecMp3;
    for (size_t i = 0; i + 1 < n; ++i)
        if (d[i] == 0xff && (d[i + 1] & 0xe0) == 0xe0) return ((d[i + 1] >> 1) & 3) == 0 ? kCodecAac : kCodecMp3;
    return n >= 8192 ? kCodecAac : 0;  // no sync in 8 KiB: let the ADTS path resynchronise (it ends the stream at EOF)
}

// An MP3 stream's pass (nanomp3 inside Mp3Decoder::decode_i16, soundkit-mp3/src/lib.rs:279-305, up to its Huffman stage): frame
// sync, header, side information, bit reservoir, scale factors and Huffman codes on this thread; what is left -- requantisation,
// stereo, reorder, hybrid synthesis -- are the granules handed to the tick.  A frame that cannot be decoded (the reservoir
// does not reach back far enough, damaged Huffman data, a feature not built) is consumed without output, as sk_mp3_decoder_*
// does.  At most `limit` granules per pass.
void parse_some_mp3(sk_lane *p, PStream &s, uint32_t limit, Parsed &r) {
    r.mp3 = true;
    auto fail = [&](int32_t st, const std::string &msg) {
        r.failed = true;
        r.fail_status = st;
        r.fail_msg = msg;
    };
    int rc = ensure_mp3(p);
    if (rc != SK_OK) {
        fail(rc, std::string("Decoding failed: MP3 tables: ") + sk_strerror(rc));
        return;
    }
    std::vector<sk_mp3_frame_info> found(64);
    std::vector<uint8_t> main;
    constexpr size_t kReservoirKept = 2048;
    while (r.n_frames < limit && !r.failed) {
        const size_t avail = s.pending.size() - s.pending_pos;
        uint32_t n_found = 0;
        size_t scanned = 0;
        // the free-format length: s.mp3_free_format is the one in force at pending_pos (behind the last frame taken); the scan's own
        // end state replaces it only when everything scanned is consumed
        uint32_t free_format_scan = s.mp3_free_format;
        if (avail >= 4) {
            free_format_scan = s.mp3_free_format;
            rc = sk_mp3_scan_free(s.pending.data() + s.pending_pos, avail, found.data(), (uint32_t)found.size(), &n_found, &scanned, &free_format_scan);
            if (rc != SK_OK) {
                fail(rc, std::string("Decoding failed: ") + sk_strerror(rc));
                break;
            }
            if (n_found > found.size()) n_found = (uint32_t)found.size();
        }
        if (n_found == 0) {
            s.pending_pos += scanned;  // garbage in front of an incomplete frame goes
            s.mp3_free_format = free_format_scan;
            if (s.saw_eof) {
                r.eof = true;
                break;
            }
            const int got = pull_input(s);
            if (got == 0) break;  // needs more input
            continue;
        }
        size_t consumed = 0;
        for (uint32_t k = 0; k < n_found && !r.failed; ++k) {
            const sk_mp3_frame_info &h = found[k];
            if (r.n_frames + h.granules > limit) {
                r.budget_stop = true;
                break;
            }
            const uint8_t *frame = s.pending.data() + s.pending_pos + h.offset;
            const size_t head = 4u + (h.has_crc ? 2u : 0u) + h.side_info_bytes;
            sk_mp3_side_info side;
            bool ok = sk_mp3_parse_side_info(frame, h.frame_bytes, &h, &side) == SK_OK;
            size_t main_len = 0;
            main.resize(s.mp3_reservoir.size() + h.frame_bytes);
            if (ok) ok = sk_mp3_main_data(frame, h.frame_bytes, &h, &side, s.mp3_reservoir.data(), s.mp3_reservoir.size(), main.data(), main.size(), &main_len) == SK_OK;
            sk_mp3_granule_data data[2][2];
            if (ok) ok = sk_mp3_decode_main_data(p->mp3_cb, &h, &side, main.data(), main_len, data) == SK_OK;
            const bool joint = h.mode == 1;
            if (ok) {
                if (s.engine_stream == kNoStream) {  // the first frame that decodes fixes rate and channels (lib.rs:203-204)
                    s.rate = h.sample_rate;
                    s.channels = h.channels;
                    rc = sk_stream_open(p->engine, s.rate, s.channels, &s.engine_stream);
                    if (rc != SK_OK) {
                        fail(rc, std::string("Decoding failed: engine stream: ") + sk_strerror(rc));
                        break;
                    }
                    const uint32_t target = s.opt.output_sample_rate ? s.opt.output_sample_rate : s.rate;
                    s.resample = target != s.rate;
                    if (s.resample) {
                        rc = sk_resampler_open(p->engine, s.engine_stream, s.rate, target);
                        if (rc != SK_OK) {
                            fail(rc, "Decoding failed: Failed to create resampler: unsupported rate pair");
                            break;
                        }
                    }
                } else if (h.channels != s.channels || h.sample_rate != s.rate) {
                    fail(SK_MP3_UNSUPPORTED, "Decoding failed: MP3 sample rate or channel count changed mid-stream");
                    break;
                }
                for (int gr = 0; gr < h.granules; ++gr) {
                    sk_mp3_requant_granule g;
                    std::memset(&g, 0, sizeof g);
                    g.sample_rate = h.sample_rate;
                    g.channels = h.channels;
                    g.ms_stereo = joint && (h.mode_ext & 2);
                    g.intensity_stereo = joint && (h.mode_ext & 1);
                    g.lsf = h.version != 1;
                    if (g.lsf && g.intensity_stereo && h.channels == 2 && data[gr][1].intensity_scale) g.intensity_stereo |= 2;
                    sk_mp3_granule_desc d;
                    std::memset(&d, 0, sizeof d);
                    d.stream = s.engine_stream;
                    d.channels = h.channels;
                    for (int ch = 0; ch < h.channels; ++ch) {
                        const sk_mp3_granule_side &gs = side.gr[gr][ch];
                        const sk_mp3_granule_data &src = data[gr][ch];
                        sk_mp3_requant_channel &c = g.ch[ch];
                        c.global_gain = gs.global_gain, c.scalefac_scale = gs.scalefac_scale, c.preflag = src.preflag;
                        c.block_type = gs.block_type, c.mixed_block_flag = gs.mixed_block_flag;
                        std::memcpy(c.subblock_gain, gs.subblock_gain, 3);
                        std::memcpy(c.scalefac_l, src.scalefac_l, 22);
                        std::memcpy(c.scalefac_s, src.scalefac_s, 39);
                        d.block_type[ch] = gs.block_type, d.mixed_block_flag[ch] = gs.mixed_block_flag;
                        r.mp3_is.insert(r.mp3_is.end(), src.is, src.is + 576);
                    }
                    r.mp3_gr.push_back(g);
                    r.mp3_desc.push_back(d);
                    r.n_frames += 1;
                }
            }
            // whatever became of the frame, its own main data is what later frames reach back into
            if (h.frame_bytes > head) s.mp3_reservoir.insert(s.mp3_reservoir.end(), frame + head, frame + h.frame_bytes);
            if (s.mp3_reservoir.size() > 4 * kReservoirKept) s.mp3_reservoir.erase(s.mp3_reservoir.begin(), s.mp3_reservoir.end() - (ptrdiff_t)kReservoirKept);
            consumed = h.offset + h.frame_bytes;
            if ((frame[2] >> 4) == 0) s.mp3_free_format = h.frame_bytes - h.padding;
            if (k + 1 == n_found && n_found < found.size()) consumed = scanned, s.mp3_free_format = free_format_scan;
        }
        s.pending_pos += consumed;
        if (r.budget_stop) break;
    }
}

// Pulls ADTS frames out of the stream's byte queue and runs the front-end on them, at most `limit` frames.
void parse_some(sk_lane *p, PStream &s, uint32_t limit, float *coeffs, sk_aac_frame_desc *descs, std::vector<uint8_t> &au_stage,
                sk_au_item *au_items, Parsed &r) {
    const bool gpu_entropy = p->cfg.gpu_entropy == 1;
    const bool quant = p->cfg.gpu_entropy == 2;  // host Huffman decode, the rest of the front-end on the device (sk_tick_run_q)
    auto fail = [&](int32_t st, const std::string &msg) {
        r.failed = true;
        r.fail_status = st;
        r.fail_msg = msg;
    };
    while (s.codec == 0) {  // the worker's detect_and_init_decoder: the stream's first bytes choose the decoder
        s.codec = sniff_codec(s.pending.data() + s.pending_pos, s.pending.size() - s.pending_pos);
        if (s.codec) break;
        if (s.saw_eof) {
            s.codec = kCodecAac;  // too short to tell: the ADTS path ends it
            break;
        }
        if (pull_input(s) == 0) return;  // needs more input
    }
    if (s.codec == kCodecMp3) {
        parse_some_mp3(p, s, limit, r);
        return;
    }
    while (r.n_frames < limit && !r.failed) {
        // make sure a whole frame is in `pending`
        size_t avail = s.pending.size() - s.pending_pos;
        size_t frame_len = 0, pay_off = 0, pay_len = 0;
        uint8_t asc[2];
        bool have = false;
        while (avail >= 7) {
            const uint8_t *d = s.pending.data() + s.pending_pos;
            if (sk_adts_parse(d, avail, &frame_len, &pay_off, &pay_len, asc) != SK_OK) {  // resynchronise
                s.pending_pos += 1;
                avail -= 1;
                continue;
            }
            have = avail >= frame_len;
            break;
        }
        if (!have) {
            if (s.saw_eof) {  // whatever is left is not a frame: flush_decoder has nothing more to give
                r.eof = true;
                break;
            }
            if (pull_input(s) == 0) break;  // needs more input
            continue;
        }
        const uint8_t *frame = s.pending.data() + s.pending_pos;
        if (!s.fe) {  // first header: AacLcDecoder::from_audio_specific_config (lib.rs:1007-1027, decoder.rs:80)
            int rc = sk_aac_decoder_create(asc, 2, &s.fe);
            if (rc != SK_OK) {
                fail(rc, std::string("Decoding failed: AAC-LC decoder init: ") + sk_strerror(rc));
                break;
            }
            s.asc[0] = asc[0];
            s.asc[1] = asc[1];
            (void)sk_aac_decoder_info(s.fe, &s.rate, &s.channels);
            rc = sk_stream_open(p->engine, s.rate, s.channels, &s.engine_stream);
            if (rc != SK_OK) {
                fail(rc, std::string("Decoding failed: engine stream: ") + sk_strerror(rc));
                break;
            }
            const uint32_t target = s.opt.output_sample_rate ? s.opt.output_sample_rate : s.rate;
            s.resample = target != s.rate;
            if (s.resample) {
                rc = sk_resampler_open(p->engine, s.engine_stream, s.rate, target);
                if (rc != SK_OK) {  // StreamingResampler::new failing, lib.rs:3405-3417
                    fail(rc, "Decoding failed: Failed to create resampler: unsupported rate pair");
                    break;
                }
            }
        } else if (asc[0] != s.asc[0] || asc[1] != s.asc[1]) {
            fail(SK_AAC_ERR_UNSUPPORTED_FEATURE, "Decoding failed: AAC configuration changed mid-stream");
            break;
        }
        if (gpu_entropy) {  // framing only: the access unit itself goes to the device
            const size_t padded = (pay_len + 8 + 3) & ~(size_t)3;
            // one pass never stages more than an empty batch holds (a stream of maximum-length ADTS frames would otherwise
            // wait for room that cannot come); the rest of the stream's frames go into the next tick
            if (r.n_frames > 0 && r.n_au_bytes + padded > p->au_pass_budget) {
                r.budget_stop = true;
                break;
            }
            if (au_stage.size() < r.n_au_bytes + padded) au_stage.resize(r.n_au_bytes + padded + 4096);
            std::memcpy(au_stage.data() + r.n_au_bytes, frame + pay_off, pay_len);
            std::memset(au_stage.data() + r.n_au_bytes + pay_len, 0, padded - pay_len);
            au_items[r.n_frames] = sk_au_item{(uint32_t)r.n_au_bytes, (uint32_t)pay_len};
            r.n_au_bytes += padded;
        } else if (quant) {  // i16 values where the f32 spectra would go, side records staged like access units
            sk_aac_frame_desc &d = descs[r.n_frames];
            if (au_stage.size() < r.n_au_bytes + SK_AAC_UNIT_SIDE_BYTES) au_stage.resize(r.n_au_bytes + SK_AAC_UNIT_SIDE_BYTES + 8192);
            const int rc = sk_aac_decoder_parse_q(s.fe, frame + pay_off, pay_len, reinterpret_cast<int16_t *>(coeffs) + r.n_floats,
                                                  au_stage.data() + r.n_au_bytes, &d);
            if (rc != SK_OK) {
                fail(rc, std::string("Decoding failed: ") + sk_aac_decoder_last_error(s.fe));
                break;
            }
            d.stream = s.engine_stream;
            r.n_au_bytes += SK_AAC_UNIT_SIDE_BYTES;
            r.raw.insert(r.raw.end(), frame + pay_off, frame + pay_off + pay_len);
            r.raw_len.push_back((uint32_t)pay_len);
        } else {
            sk_aac_frame_desc &d = descs[r.n_frames];
            const int rc = sk_aac_decoder_parse(s.fe, frame + pay_off, pay_len, coeffs + r.n_floats, &d);
            if (rc != SK_OK) {
                fail(rc, std::string("Decoding failed: ") + sk_aac_decoder_last_error(s.fe));
                break;
            }
            d.stream = s.engine_stream;
        }
        s.pending_pos += frame_len;
        r.n_frames += 1;
        r.n_floats += (size_t)s.channels * 1024;
    }
}

int ensure_mp3(sk_lane *p) {
    if (p->mp3_ready.load(std::memory_order_acquire)) return SK_OK;
    std::lock_guard<std::mutex> lk(p->mp3_mu);
    if (p->mp3_ready.load(std::memory_order_relaxed)) return SK_OK;
    sk_mp3_tables t;
    int rc = sk_mp3_iso_tables(&t);
    if (rc != SK_OK) return rc;
    static const uint32_t rates[9] = {44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000};
    for (int row = 0; row < 9 && rc == SK_OK; ++row) rc = sk_mp3_set_band_tables(p->engine, rates[row], t.long_offsets[row], t.short_offsets[row], t.pretab);
    if (rc == SK_OK) rc = sk_mp3_set_synthesis_window(p->engine, t.window);
    if (rc != SK_OK) return rc;
    if (!p->mp3_cb) {
        rc = sk_mp3_codebook_create(&t, &p->mp3_cb);
        if (rc != SK_OK) return rc;
    }
    {
        // the batches' granule arrays: workers copy into disjoint ranges without the lock, so they are sized once, here, while
        // no MP3 claim can exist yet (an MP3 pass reaches the batches only behind this function)
        std::lock_guard<std::mutex> bl(p->batch_mu);
        for (Batch &b : p->batches) {
            if (b.mp3_is) continue;
            const size_t cap = p->cfg.max_frames_per_tick;
            if (hipHostMalloc((void **)&b.mp3_is, cap * 2 * 576 * sizeof(int16_t) + 64, hipHostMallocPortable) != hipSuccess) {
                b.mp3_is = nullptr;
                return SK_ERR_OOM;
            }
            b.mp3_gr.resize(cap);
            b.mp3_desc.resize(cap);
        }
    }
    p->mp3_ready.store(true, std::memory_order_release);
    return SK_OK;
}

// test hook (sk_debug_throw_in_thread): the n-th passage of point `where` throws std::bad_alloc
std::atomic<int> g_thread_throw_after{-1}, g_thread_throw_where{-1};
void thread_debug_point(int where) {
    if (g_thread_throw_after.load(std::memory_order_relaxed) < 0 || g_thread_throw_where.load(std::memory_order_relaxed) != where) return;
    if (g_thread_throw_after.fetch_sub(1) == 0) throw std::bad_alloc();
}

void worker_body(sk_lane *p) {
    const uint32_t per_stream = p->cfg.max_stream_frames_per_tick;
    const bool gpu_entropy = p->cfg.gpu_entropy == 1, quant = p->cfg.gpu_entropy == 2;
    std::vector<float> coeffs(gpu_entropy ? 0 : (size_t)per_stream * 2 * 1024);  // quant: the same storage holds i16 (half of it)
    std::vector<sk_aac_frame_desc> descs(per_stream);
    std::vector<uint8_t> au_stage;
    std::vector<sk_au_item> au_items(per_stream);
    for (;;) {
        uint32_t handle;
        {
            std::unique_lock<std::mutex> lk(p->rq_mu);
            p->workers_waiting_ready.fetch_add(1);
            p->rq_cv.wait(lk, [&] { return p->stop || !p->ready.empty(); });
            p->workers_waiting_ready.fetch_sub(1);
            if (p->stop) return;
            handle = p->ready.front();
            p->ready.pop_front();
        }
        thread_debug_point(1);
        PStream &s = *p->streams[handle];
        uint32_t room;
        {
            std::lock_guard<std::mutex> lk(s.mu);
            s.queued = false;
            if (!s.open || s.busy || s.finished || s.cancelled) continue;
            room = s.out.size() < p->cfg.output_buffer ? p->cfg.output_buffer - (uint32_t)s.out.size() : 0;
            if (room == 0) continue;
            s.busy = true;
        }
        Parsed r;
        p->workers_parsing.fetch_add(1);
        const Clock::time_point t0 = Clock::now();
        // room in the output queue bounds the access units of this pass: one AudioData per unit, or -- through the
        // streaming resampler -- one per completed chunk of 4096 source frames = 4 units (lib.rs:1970-2003)
        const uint32_t limit = std::min(per_stream, s.resample ? (room > per_stream / 4 ? per_stream : 4 * room) : room);
        try {
            thread_debug_point(0);
            parse_some(p, s, limit, coeffs.data(), descs.data(), au_stage, au_items.data(), r);
        } catch (...) {
            // whatever was thrown while this stream's input was parsed is this stream's error and nobody else's
            // (soundkit-decoder/src/lib.rs:3131-3134); the frames of this pass are dropped with it
            r = Parsed{};
            r.failed = true;
            r.fail_status = sk::abi_caught("entropy thread");
            try {
                r.fail_msg = std::string("Decoding failed: ") + sk_strerror(r.fail_status) + ": " + sk::abi_message();
            } catch (...) {
            }
        }
        {
            std::lock_guard<std::mutex> lk(s.mu);  // read by mark_schedulable and the state dump
            s.more = (r.n_frames == limit || r.budget_stop) && !r.eof && !r.failed;
        }
        p->parse_ns.fetch_add(ns_since(t0));
        p->workers_parsing.fetch_sub(1);
        if (r.n_frames == 0 && !r.eof && !r.failed) {  // nothing complete yet
            bool dropped;
            {
                std::lock_guard<std::mutex> lk(s.mu);
                s.busy = false;
                dropped = s.cancelled;
                if (dropped) s.open = false;
                maybe_schedule(p, s, handle);
            }
            if (dropped) {  // the handle was dropped while this worker held the stream
                release_device_side(p, s);
                std::lock_guard<std::mutex> lk(p->handles_mu);
                p->free_handles.push_back(handle);
            }
            continue;
        }
        // claim room in the batch being filled
        Batch *b;
        size_t desc_at, float_at, au_at, mp3_at = 0, mp3_row_at = 0;
        {
            std::unique_lock<std::mutex> lk(p->batch_mu);
            p->workers_waiting_room.fetch_add(1);
            p->room_cv.wait(lk, [&] {
                const Batch &f = p->batches[p->filling];
                return p->stop || (f.n_descs + f.n_mp3 + r.n_frames <= p->cfg.max_frames_per_tick &&
                                   (gpu_entropy ? f.au_used + r.n_au_bytes <= f.au_cap : f.n_floats + r.n_floats <= f.coeff_cap) &&
                                   (!quant || f.au_used + r.n_au_bytes <= f.au_cap));
            });
            p->workers_waiting_room.fetch_sub(1);
            if (p->stop) return;
            b = &p->batches[p->filling];
            desc_at = b->n_descs;
            float_at = b->n_floats;
            au_at = b->au_used;
            mp3_at = b->n_mp3;
            mp3_row_at = b->mp3_rows;
            if (r.mp3) {
                b->n_mp3 += r.n_frames;
                b->mp3_rows += r.mp3_is.size() / 576;
            } else {
                b->n_descs += r.n_frames;
            }
            b->n_floats += r.n_floats;
            b->au_used += r.n_au_bytes;
            sk_tick_stream t{};
            t.codec = r.mp3 ? SK_TICK_MP3 : SK_TICK_AAC;
            t.stream = s.engine_stream == kNoStream ? 0 : s.engine_stream;
            t.n_frames = r.n_frames;
            t.out_bits = s.opt.output_bits_per_sample ? s.opt.output_bits_per_sample : 16;
            t.out_channels = s.opt.output_channels ? s.opt.output_channels : s.channels;
            t.resample = s.resample ? 1 : 0;
            t.flush = (r.eof && s.resample) ? 1 : 0;
            BatchEntry be;
            be.handle = handle;
            be.eof = r.eof;
            be.failed = r.failed;
            be.fail_status = r.fail_status;
            be.fail_msg = std::move(r.fail_msg);
            be.raw = std::move(r.raw);
            be.raw_len = std::move(r.raw_len);
            if (s.engine_stream == kNoStream) {  // ended before a single header was seen: nothing for the device
                t.n_frames = 0;
                t.flush = 0;
            }
            b->ts.push_back(t);
            b->entries.push_back(std::move(be));
            b->writers += 1;
        }
        if (r.n_frames && r.mp3) {
            std::memcpy(b->mp3_gr.data() + mp3_at, r.mp3_gr.data(), r.n_frames * sizeof(sk_mp3_requant_granule));
            std::memcpy(b->mp3_desc.data() + mp3_at, r.mp3_desc.data(), r.n_frames * sizeof(sk_mp3_granule_desc));
            std::memcpy(b->mp3_is + mp3_row_at * 576, r.mp3_is.data(), r.mp3_is.size() * sizeof(int16_t));
        } else if (r.n_frames && gpu_entropy) {
            std::memcpy(b->au_bytes + au_at, au_stage.data(), r.n_au_bytes);
            for (uint32_t k = 0; k < r.n_frames; ++k)
                b->units[desc_at + k] = sk_au_item{(uint32_t)(au_at + au_items[k].byte_offset), au_items[k].byte_len};
        } else if (r.n_frames && quant) {
            std::memcpy(b->descs.data() + desc_at, descs.data(), r.n_frames * sizeof(sk_aac_frame_desc));
            std::memcpy(reinterpret_cast<int16_t *>(b->coeffs) + float_at, coeffs.data(), r.n_floats * sizeof(int16_t));
            std::memcpy(b->au_bytes + au_at, au_stage.data(), r.n_au_bytes);
        } else if (r.n_frames) {
            std::memcpy(b->descs.data() + desc_at, descs.data(), r.n_frames * sizeof(sk_aac_frame_desc));
            std::memcpy(b->coeffs + float_at, coeffs.data(), r.n_floats * sizeof(float));
        }
        {
            std::lock_guard<std::mutex> lk(p->batch_mu);
            b->writers -= 1;
        }
        p->batch_cv.notify_one();  // only the submission thread waits on it
    }
}

void push_error(PStream &s, int32_t status, const std::string &msg) {
    Output o;
    o.is_error = true;
    o.status = status;
    o.data.assign(msg.begin(), msg.end());
    s.out.push_back(std::move(o));
}

void submit_body(sk_lane *p) {
    std::vector<sk_tick_stream> ts;
    for (;;) {
        Batch *b;
        int index;
        {
            std::unique_lock<std::mutex> lk(p->batch_mu);
            const Clock::time_point t_idle = Clock::now();
            p->submit_where = 0;
            p->batch_cv.wait(lk, [&] { return p->stop || !p->batches[p->filling].ts.empty(); });
            if (p->stop) return;
            p->submit_where = 1;
            // let the batch fill for a moment unless it is already full
            // (system_clock deadline: pthread_cond_timedwait, which every sanitizer runtime understands)
            const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(p->cfg.tick_wait_us);
            p->batch_cv.wait_until(lk, deadline, [&] {
                return p->stop || p->batches[p->filling].n_descs + p->cfg.max_stream_frames_per_tick > p->cfg.max_frames_per_tick;
            });
            // the workers move on to a free batch while this one runs
            p->submit_where = 2;
            p->batch_cv.wait(lk, [&] { return p->stop || (p->batches[p->filling].writers == 0 && !p->free_batches.empty()); });
            if (p->stop) return;
            p->submit_where = 3;
            p->idle_ns.fetch_add(ns_since(t_idle));
            index = p->filling;
            b = &p->batches[index];
            p->filling = p->free_batches.front();
            p->free_batches.pop_front();
        }
        p->room_cv.notify_all();  // room again
        thread_debug_point(2);

        const Clock::time_point t0 = Clock::now();
        const uint32_t n_streams = (uint32_t)b->ts.size(), n_frames = (uint32_t)b->n_descs;
        // a stream may have ended without ever reaching the device: it gets no row in the tick's table
        ts.clear();
        b->row_of.clear();
        b->entry_row.assign(n_streams, kNoStream);
        for (uint32_t i = 0; i < n_streams; ++i) {
            PStream &s = *p->streams[b->entries[i].handle];
            if (s.engine_stream == kNoStream) continue;
            ts.push_back(b->ts[i]);
            b->entry_row[i] = (uint32_t)b->row_of.size();
            b->row_of.push_back(i);
        }
        uint32_t max_out = 0;
        size_t used = 0;
        b->rc = SK_OK;
        b->n_out = 0;
        if (!ts.empty()) {
            const size_t bound = sk_tick_out_bound_on(p->engine, ts.data(), (uint32_t)ts.size(), &max_out);
            if (bound > b->out_pinned_cap) {
                // Pinned memory is slow to get (10 ms per 64 MB on an idle device, many times that beside a running lane): the
                // bound is the engine's own (the streams' real ratios and widths, not the 8 -> 48 kHz stereo 32-bit case), with room
                // to grow into, and what one batch of the lane has needed the others ask for at once.
                const Clock::time_point t_alloc = Clock::now();
                const size_t want = std::max(bound + bound / 2, p->out_cap_seen.load());
                if (b->out_pinned) (void)hipHostFree(b->out_pinned);
                b->out_pinned = nullptr;
                b->out_pinned_cap = 0;
                if (hipHostMalloc((void **)&b->out_pinned, want, hipHostMallocPortable) == hipSuccess) {
                    b->out_pinned_cap = want;
                    p->out_cap_seen.store(want);
                } else {
                    b->rc = SK_ERR_OOM;
                }
                static const bool trace = std::getenv("SK_TICK_TRACE") != nullptr;
                if (trace) std::fprintf(stderr, "sk_pipeline: output buffer of batch %d regrown to %zu bytes in %.2f ms\n", index, b->out_pinned_cap, ns_since(t_alloc) * 1e-6);
            }
            if (b->recs.size() < max_out) b->recs.resize(max_out);
            if (b->rc == SK_OK && b->n_mp3) {  // streams of both codecs in this tick: the AAC units in the lane's form, the MP3 granules beside them
                sk_tick_input in{};
                in.n_aac_units = n_frames;
                if (p->cfg.gpu_entropy == 2) in.descs = b->descs.data(), in.q_sides = b->au_bytes, in.q_quant = reinterpret_cast<const int16_t *>(b->coeffs);
                else if (p->cfg.gpu_entropy) in.units = b->units.data(), in.au_bytes = b->au_bytes, in.au_bytes_len = b->au_used + 8;
                else in.descs = b->descs.data(), in.coeffs = b->coeffs;
                if (n_frames == 0) in.descs = nullptr, in.coeffs = nullptr, in.units = nullptr, in.q_sides = nullptr;
                in.n_mp3_granules = (uint32_t)b->n_mp3;
                in.mp3_granules = b->mp3_gr.data(), in.mp3_descs = b->mp3_desc.data(), in.mp3_is = b->mp3_is;
                b->rc = sk_tick_run_mixed(p->engine, ts.data(), (uint32_t)ts.size(), &in, b->out_pinned, b->out_pinned_cap, b->recs.data(), max_out,
                                          &b->n_out, &used);
            } else if (b->rc == SK_OK && p->cfg.gpu_entropy == 2)
                b->rc = sk_tick_run_q(p->engine, ts.data(), (uint32_t)ts.size(), b->descs.data(), b->au_bytes,
                                      reinterpret_cast<const int16_t *>(b->coeffs), n_frames, b->out_pinned, b->out_pinned_cap, b->recs.data(),
                                      max_out, &b->n_out, &used);
            else if (b->rc == SK_OK && p->cfg.gpu_entropy)
                b->rc = sk_tick_run_au(p->engine, ts.data(), (uint32_t)ts.size(), b->units.data(), n_frames, b->au_bytes,
                                       b->au_used + 8, b->out_pinned, b->out_pinned_cap, b->recs.data(), max_out, &b->n_out, &used);
            else if (b->rc == SK_OK)
                b->rc = sk_tick_run(p->engine, ts.data(), (uint32_t)ts.size(), b->descs.data(), b->coeffs, n_frames, b->out_pinned,
                                    b->out_pinned_cap, b->recs.data(), max_out, &b->n_out, &used);
        }
        p->submit_where = 4;
        p->tick_ns.fetch_add(ns_since(t0));
        p->n_ticks.fetch_add(1);
        p->n_frames.fetch_add(n_frames + (uint32_t)b->n_mp3);
        b->rec_begin.assign(b->row_of.size() + 1, b->n_out);
        {
            uint32_t k = 0;
            for (uint32_t row = 0; row < b->row_of.size(); ++row) {
                b->rec_begin[row] = k;
                while (b->rc == SK_OK && k < b->n_out && b->recs[k].stream_index == row) ++k;
            }
        }
        {
            std::lock_guard<std::mutex> lk(p->batch_mu);
            for (uint32_t d = 0; d < p->n_deliver; ++d) p->to_deliver.push_back(index);  // one slice per delivery thread
        }
        p->deliver_cv.notify_all();
    }
}

// The message behind a status the device reported for a unit of this entry.  The device reports codes; the reference's
// text comes from parsing the entry's access units of this tick once more on the host (errors are rare, and an entry holds
// at most max_stream_frames_per_tick units).
std::string device_error_text(sk_lane *p, const Batch *b, uint32_t entry, const PStream &s, int32_t status) {
    std::string msg = "Decoding failed: invalid AAC config: frame rejected by the synthesis engine";
    if (!(p->cfg.gpu_entropy && status <= -101 && status >= -108)) return msg;
    msg = std::string("Decoding failed: ") + sk_strerror(status);  // found on the device (stereo tools, TNS) or in the unit's tail
    sk_aac_decoder *probe = nullptr;
    if (sk_aac_decoder_create(s.asc, 2, &probe) != SK_OK) return msg;
    std::vector<float> sink(2048);
    sk_aac_frame_desc d;
    if (p->cfg.gpu_entropy == 1) {
        size_t first = 0;
        for (uint32_t q = 0; q < entry; ++q) first += b->ts[q].n_frames;
        for (uint32_t u = 0; u < b->ts[entry].n_frames; ++u) {
            const sk_au_item &it = b->units[first + u];
            if (sk_aac_decoder_parse(probe, b->au_bytes + it.byte_offset, it.byte_len, sink.data(), &d) != SK_OK) {
                msg = std::string("Decoding failed: ") + sk_aac_decoder_last_error(probe);
                break;
            }
        }
    } else {  // quantised hand-over: the worker kept the pass's access units beside the entry for exactly this
        const BatchEntry &be = b->entries[entry];
        size_t at = 0;
        for (uint32_t len : be.raw_len) {
            if (sk_aac_decoder_parse(probe, be.raw.data() + at, len, sink.data(), &d) != SK_OK) {
                msg = std::string("Decoding failed: ") + sk_aac_decoder_last_error(probe);
                break;
            }
            at += len;
        }
    }
    sk_aac_decoder_destroy(probe);
    return msg;
}

void watchdog_main(sk_lane *p, int secs);

// Hands a finished tick's outputs to the streams' queues: outputs first (in order), then the end-of-stream /
// error notes, then the stream is free to be parsed again.
void deliver_body(sk_lane *p) {
    std::vector<uint32_t> wake;  // streams that can be parsed again: queued in one go, one wake-up
    std::vector<uint32_t> listed;  // handles that now have something to receive
    for (;;) {
        Batch *b;
        int index;
        uint32_t slice;
        {
            std::unique_lock<std::mutex> lk(p->batch_mu);
            p->deliverers_waiting.fetch_add(1);
            p->deliver_cv.wait(lk, [&] { return p->stop || !p->to_deliver.empty(); });
            p->deliverers_waiting.fetch_sub(1);
            if (p->stop) return;
            index = p->to_deliver.front();
            p->to_deliver.pop_front();
            b = &p->batches[index];
            slice = b->next_slice++;  // this thread serves the rows / entries congruent to `slice`
        }
        thread_debug_point(3);
        const uint32_t n_slices = p->n_deliver;
        const Clock::time_point t_deliver = Clock::now();
        const uint32_t n_streams = (uint32_t)b->ts.size();
        const int rc = b->rc;
        wake.clear();
        listed.clear();
        // One delivery thread serves an entry completely -- its outputs (the records of its tick row are contiguous), then
        // the end-of-stream / error note, then the stream is free again -- under one hold of the stream's lock.  (Rows and
        // entries used to be sliced separately: with several delivery threads a stream without a tick row in front of
        // another shifted the two numberings against each other, and that stream's end could be announced by one thread
        // before another had queued its last outputs.)
        for (uint32_t i = slice; i < n_streams; i += n_slices) {
            BatchEntry &be = b->entries[i];
            PStream &s = *p->streams[be.handle];
            const uint32_t row = b->entry_row[i];
            bool release = false, was_cancelled = false;  // was_cancelled: as seen under the lock that cleared `busy` -- a cancel that
                                                          // arrives later finds the stream idle and frees the slot itself
            {
                std::lock_guard<std::mutex> lk(s.mu);
                for (uint32_t k = row == kNoStream ? 0 : b->rec_begin[row]; row != kNoStream && rc == SK_OK && k < b->rec_begin[row + 1]; ++k) {
                    const sk_tick_output &r = b->recs[k];
                    if (s.cancelled || s.finished) continue;
                    if (r.status != 0) {
                        push_error(s, r.status, device_error_text(p, b, i, s, r.status));
                        s.finished = true;
                        p->n_errors.fetch_add(1);
                        continue;
                    }
                    Output o;
                    o.rate = s.opt.output_sample_rate ? s.opt.output_sample_rate : s.rate;
                    o.frames = r.frames;
                    o.bits = r.bits;
                    o.channels = r.channels;
                    o.data.assign(b->out_pinned + r.byte_offset, b->out_pinned + r.byte_offset + r.bytes);
                    s.out.push_back(std::move(o));
                    p->n_outputs.fetch_add(1);
                }
                if (rc != SK_OK && !s.finished) {
                    push_error(s, rc, std::string("Decoding failed: engine tick: ") + sk_strerror(rc));
                    s.finished = true;
                    p->n_errors.fetch_add(1);
                } else if (be.failed && !s.finished) {
                    push_error(s, be.fail_status, be.fail_msg);
                    s.finished = true;
                    p->n_errors.fetch_add(1);
                } else if (be.eof) {
                    s.finished = true;
                }
                s.busy = false;
                was_cancelled = s.cancelled;
                release = s.finished || was_cancelled;
                if (s.cancelled) {
                    s.out.clear();
                    s.in.clear();
                }
                if (!release && mark_schedulable(p, s)) wake.push_back(be.handle);
                if (!s.cancelled && !s.out_listed && (!s.out.empty() || s.finished)) {
                    s.out_listed = true;
                    listed.push_back(be.handle);
                }
                s.cv_out.notify_all();
            }
            if (release) {
                release_device_side(p, s);
                if (was_cancelled) {  // the handle was dropped while its frames were in flight: free the slot now
                    {
                        std::lock_guard<std::mutex> lk(s.mu);
                        s.open = false;
                    }
                    std::lock_guard<std::mutex> lk(p->handles_mu);
                    p->free_handles.push_back(be.handle);
                }
            }
        }
        if (!wake.empty()) {
            {
                std::lock_guard<std::mutex> lk(p->rq_mu);
                p->ready.insert(p->ready.end(), wake.begin(), wake.end());
            }
            p->rq_cv.notify_all();
        }
        if (!listed.empty()) {
            {
                std::lock_guard<std::mutex> lk(p->oq->mu);
                for (uint32_t h : listed) p->oq->ready.push_back(h * p->n_lanes + p->lane_index);
            }
            p->oq->cv.notify_all();
        }
        bool last;
        {
            std::lock_guard<std::mutex> lk(p->batch_mu);
            last = ++b->slices_done == n_slices;
            if (last) {
                b->clear();
                p->free_batches.push_back(index);
            }
        }
        if (last) p->batch_cv.notify_one();  // the submission thread may be waiting for a free batch
        p->deliver_ns.fetch_add(ns_since(t_deliver));
    }
}

// A thread of the lane that dies of an exception outside the per-stream guard (an allocation while it held a batch, say)
// cannot know which invariants it left broken, so the lane stops as a whole -- but it stops as an ERROR, not as
// std::terminate: every open stream gets the status as its last output (the reference's "an error ends the stream",
// soundkit-decoder/src/lib.rs:3131-3134, for all of the lane's streams at once), later calls on the lane return it, and the
// process lives.  Never throws.
void lane_fatal(sk_lane *p, int status) noexcept {
    int expected = 0;
    if (!p->fatal.compare_exchange_strong(expected, status)) return;  // once
    char text[320];
    std::snprintf(text, sizeof text, "Decoding failed: %s: %s", sk_strerror(status), sk::abi_message());
    for (uint32_t h = 0; h < p->streams.size(); ++h) {
        PStream &s = *p->streams[h];
        bool list = false;
        try {
            std::lock_guard<std::mutex> lk(s.mu);
            if (!s.open || s.cancelled) continue;
            if (!s.finished) {
                try {
                    push_error(s, status, text);
                } catch (...) {
                }
                s.finished = true;
                p->n_errors.fetch_add(1);
            }
            // s.busy stays: a thread that has not noticed the stop yet may still hold the stream; a cancel of such a handle
            // defers the release, and the lane's teardown (lane_destroy, after joining the threads) releases what is left
            if (!s.out_listed) s.out_listed = list = true;
            s.cv_out.notify_all();
        } catch (...) {
        }
        if (list) {
            try {
                std::lock_guard<std::mutex> lk(p->oq->mu);
                p->oq->ready.push_back(h * p->n_lanes + p->lane_index);
            } catch (...) {
            }
            p->oq->cv.notify_all();
        }
    }
    {
        std::lock_guard<std::mutex> a(p->rq_mu);  // lane_destroy's order
        std::lock_guard<std::mutex> b(p->batch_mu);
        p->stop = true;
    }
    p->rq_cv.notify_all();
    p->batch_cv.notify_all();
    p->room_cv.notify_all();
    p->deliver_cv.notify_all();
}

template <void (*Body)(sk_lane *)>
void guarded_main(sk_lane *p, const char *what) noexcept {
    try {
        Body(p);
    } catch (...) {
        lane_fatal(p, sk::abi_caught(what));
    }
}
void worker_main(sk_lane *p) { guarded_main<worker_body>(p, "entropy thread"); }
void submit_main(sk_lane *p) { guarded_main<submit_body>(p, "submission thread"); }
void deliver_main(sk_lane *p) { guarded_main<deliver_body>(p, "delivery thread"); }

// CPUs this process may actually use: the affinity mask, cut down by a cgroup v2 / v1 CPU quota when there is one
unsigned usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (unsigned)CPU_COUNT(&set);
    double quota = 0.0;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32];
        long period = 0;
        if (std::fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) quota = std::atof(q) / (double)period;
        std::fclose(f);
    } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long q = 0, period = 100000;
        if (std::fscanf(g, "%ld", &q) != 1) q = 0;
        std::fclose(g);
        if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(h, "%ld", &period) != 1) period = 100000;
            std::fclose(h);
        }
        if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota >= 1.0 && quota < (double)n) n = (unsigned)quota;
    return n ? n : 1;
}

PStream *stream_of(sk_lane *p, uint32_t handle) {
    if (!p || handle >= p->streams.size()) return nullptr;
    return p->streams[handle].get();
}

}  // namespace

namespace {

int lane_create(sk_engine *e, const sk_pipeline_config *cfg, OutQueue *oq, uint32_t lane_index, uint32_t n_lanes, sk_lane **out) {
    if (!e || !out) return SK_ERR_INVALID_ARG;
    *out = nullptr;
    sk_lane *p = new (std::nothrow) sk_lane();
    if (!p) return SK_ERR_OOM;
    p->engine = e;
    p->oq = oq;
    p->lane_index = lane_index;
    p->n_lanes = n_lanes;
    if (cfg) p->cfg = *cfg;
    if (!p->cfg.entropy_threads) {
        const unsigned cpus = usable_cpus();  // leave room for the submission + delivery threads and the callers' own
        p->cfg.entropy_threads = cpus > 5 ? std::min(cpus - 5, 64u) : 1;
    }
    if (!p->cfg.max_streams) p->cfg.max_streams = 1024;
    // With the front-end on the GPU every access unit is a lane of its own and a tick costs about the same whatever its
    // size (the time of one unit), so ticks should be big; with the host front-end the tick is PCIe-bound and the host
    // threads want their results back soon.
    if (!p->cfg.max_frames_per_tick) p->cfg.max_frames_per_tick = p->cfg.gpu_entropy == 1 ? 65536 : 16384;
    if (!p->cfg.max_stream_frames_per_tick) p->cfg.max_stream_frames_per_tick = p->cfg.gpu_entropy == 1 ? 32 : 8;
    if (p->cfg.max_stream_frames_per_tick > p->cfg.max_frames_per_tick) p->cfg.max_stream_frames_per_tick = p->cfg.max_frames_per_tick;
    if (!p->cfg.input_buffer) p->cfg.input_buffer = 128;   // DEFAULT_INPUT_BUFFER, lib.rs:77
    if (!p->cfg.output_buffer) p->cfg.output_buffer = 16;  // DEFAULT_OUTPUT_BUFFER, lib.rs:78
    if (!p->cfg.tick_wait_us) p->cfg.tick_wait_us = p->cfg.gpu_entropy == 1 ? 2000 : 200;
    if (hipSetDevice(sk_engine_device(e)) != hipSuccess) {
        delete p;
        return SK_ERR_NO_DEVICE;
    }
    for (Batch &b : p->batches) {
        b.descs.resize(p->cfg.max_frames_per_tick);
        b.units.resize(p->cfg.max_frames_per_tick);
        hipError_t he;
        if (p->cfg.gpu_entropy == 2) {  // quantised hand-over: i16 values in the spectra's place (half of it) + one side record per unit
            b.coeff_cap = (size_t)p->cfg.max_frames_per_tick * 2 * 1024;
            b.au_cap = (size_t)p->cfg.max_frames_per_tick * SK_AAC_UNIT_SIDE_BYTES + 64;
            he = hipHostMalloc((void **)&b.coeffs, b.coeff_cap * sizeof(int16_t), hipHostMallocPortable);
            if (he == hipSuccess) he = hipHostMalloc((void **)&b.au_bytes, b.au_cap + 64, hipHostMallocPortable);
        } else if (p->cfg.gpu_entropy) {  // access units are ~0.4-0.8 KiB; 2 KiB each on average leaves room for any legal mix
            // ... and never less than one maximum-length ADTS frame (8191 bytes, padded) so that any single unit fits
            b.au_cap = std::max<size_t>((size_t)p->cfg.max_frames_per_tick * 2048, 8192 + 16) + 16384;
            p->au_pass_budget = b.au_cap;
            he = hipHostMalloc((void **)&b.au_bytes, b.au_cap + 64, hipHostMallocPortable);
        } else {
            b.coeff_cap = (size_t)p->cfg.max_frames_per_tick * 2 * 1024;
            he = hipHostMalloc((void **)&b.coeffs, b.coeff_cap * sizeof(float), hipHostMallocPortable);
        }
        if (he != hipSuccess) {
            for (Batch &x : p->batches) {
                if (x.coeffs) (void)hipHostFree(x.coeffs);
                if (x.au_bytes) (void)hipHostFree(x.au_bytes);
            }
            delete p;
            return SK_ERR_OOM;
        }
        if (b.au_bytes) std::memset(b.au_bytes, 0, b.au_cap + 64);
    }
    p->streams.resize(p->cfg.max_streams);
    for (uint32_t i = 0; i < p->cfg.max_streams; ++i) {
        p->streams[i].reset(new PStream());
        p->free_handles.push_back(p->cfg.max_streams - 1 - i);
    }
    for (int i = 1; i < sk_lane::kBatches; ++i) p->free_batches.push_back(i);
    for (uint32_t i = 0; i < p->cfg.entropy_threads; ++i) p->workers.emplace_back(worker_main, p);
    p->submitter = std::thread(submit_main, p);
    // With the front-end on the GPU the entropy threads have little to do and delivery is the busiest host stage -- and on the streams'
    // way back into the next tick: a stream is schedulable again when its outputs have been handed over.  As many delivery threads as
    // the lane has entropy threads, four at most (whole decode, 4096 streams on two lanes of four entropy threads each: 6.4 / 10.2 /
    // 10.7 / 11.3 / 11.7 M access units/s with 1 / 2 / 3 / 4 / 6 delivery threads per lane, gpurun_out/r4_ab_deliver.log).
    // With the front-end on host threads one delivery thread per three of them, three at most: MP3 (576-sample granules: many small
    // AudioData) is bound by a single delivery thread -- 2048 streams, nine entropy threads: 1.1-1.5 / 1.7-1.8 / 2.2-2.3 M granules/s
    // with 1 / 2 / 3, the single thread 0.8-0.99 busy and the run anywhere between 0.8 and 2.3 M from box to box; AAC with the host
    // front-end does not care (2.6-2.8 M access units/s with any of them; gpurun_ab/ab_mp3_deliver.sh, profiles/r04_tick_sections.md).
    p->n_deliver = p->cfg.gpu_entropy == 1 ? std::max(1u, std::min(4u, p->cfg.entropy_threads)) : std::max(1u, std::min(3u, p->cfg.entropy_threads / 3));
    if (const char *env = std::getenv("SK_PIPELINE_DELIVER_THREADS")) {  // tuning / test override
        const int n = std::atoi(env);
        if (n >= 1 && n <= 16) p->n_deliver = (uint32_t)n;
    }
    for (uint32_t d = 0; d < p->n_deliver; ++d) p->deliverers.emplace_back(deliver_main, p);
    if (const char *env = std::getenv("SK_PIPELINE_WATCHDOG")) {
        const int secs = std::atoi(env);
        if (secs > 0) p->watchdog = std::thread(watchdog_main, p, secs);
    }
    *out = p;
    return SK_OK;
}

// Where every thread of a lane stands and what the stream table looks like, as text.  Takes no lock it could wait for: the
// stream records and the queues are read under try_lock (a dump of a stuck pipeline must not get stuck itself, and the
// deques must not be read while a worker pushes to them); what could not be locked is counted as "held".
size_t lane_dump(sk_lane *p, char *buf, size_t cap) {
    size_t at = 0;
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wformat-security"  // every format below is a literal of this function
    auto put = [&](const char *fmt, auto... args) {
        if (at >= cap) return;
        const int n = std::snprintf(buf + at, cap - at, fmt, args...);
        if (n > 0) at += std::min<size_t>((size_t)n, cap - at - 1);
    };
#pragma GCC diagnostic pop
    size_t with_input = 0, busy = 0, queued = 0, out_full = 0, open = 0, more = 0, finished = 0, held = 0, with_output = 0, listed = 0,
           schedulable = 0;
    for (auto &sp : p->streams) {
        PStream &s = *sp;
        if (!s.mu.try_lock()) {
            ++held;
            continue;
        }
        if (s.open) {
            ++open;
            with_input += !s.in.empty();
            busy += s.busy;
            queued += s.queued;
            more += s.more && !s.busy;
            finished += s.finished;
            out_full += s.out.size() >= p->cfg.output_buffer;
            with_output += !s.out.empty();
            listed += s.out_listed;
            // would be scheduled if anybody asked: a non-zero count on a quiet pipeline is a lost wake-up
            schedulable += !s.busy && !s.queued && !s.finished && !s.cancelled && (!s.in.empty() || s.more) && s.out.size() < p->cfg.output_buffer;
        }
        s.mu.unlock();
    }
    put("[sk_pipeline] lane %u mode %u: ticks %llu frames %llu outputs %llu errors %llu | submitter at %d "
        "(0 idle, 1 filling, 2 waits for writers / a free batch, 3 in the tick, 4 hand-over) | workers: waiting for a stream %u, "
        "parsing %u, waiting for batch room %u of %zu | deliverers waiting %u of %u\n",
        p->lane_index, p->cfg.gpu_entropy, (unsigned long long)p->n_ticks.load(), (unsigned long long)p->n_frames.load(),
        (unsigned long long)p->n_outputs.load(), (unsigned long long)p->n_errors.load(), p->submit_where.load(),
        p->workers_waiting_ready.load(), p->workers_parsing.load(), p->workers_waiting_room.load(), p->workers.size(),
        p->deliverers_waiting.load(), p->n_deliver);
    put("    streams: open %zu (locked by others %zu) | input queued %zu | held by a worker or a tick %zu | in the ready queue %zu | "
        "stopped at the frame limit %zu | finished %zu | outputs waiting %zu, at the bound %zu, listed for wait_outputs %zu | "
        "schedulable but not queued %zu\n",
        open, held, with_input, busy, queued, more, finished, with_output, out_full, listed, schedulable);
    if (p->batch_mu.try_lock()) {
        put("    batches: filling %d, free %zu, to deliver %zu |", p->filling, p->free_batches.size(), p->to_deliver.size());
        for (int i = 0; i < sk_lane::kBatches; ++i)
            put(" [%d] streams %zu units %zu writers %u slices %u/%u |", i, p->batches[i].ts.size(), p->batches[i].n_descs, p->batches[i].writers,
                p->batches[i].slices_done, p->batches[i].next_slice);
        put("%s", "\n");
        p->batch_mu.unlock();
    } else {
        put("    batch_mu is held\n");
    }
    if (p->rq_mu.try_lock()) {
        put("    ready queue %zu\n", p->ready.size());
        p->rq_mu.unlock();
    } else {
        put("    rq_mu is held\n");
    }
    if (p->oq->mu.try_lock()) {
        put("    completion queue %zu\n", p->oq->ready.size());
        p->oq->mu.unlock();
    } else {
        put("    completion queue mutex is held\n");
    }
    const char *where = sk_engine_where(p->engine);
    put("    engine: %s\n", where ? where : "?");
    return at;
}

// SK_PIPELINE_WATCHDOG=<seconds>: prints lane_dump when no tick has finished for that long
void watchdog_main(sk_lane *p, int secs) {
    uint64_t last = p->n_ticks.load();
    int quiet = 0;
    std::vector<char> text(8192);
    for (;;) {
        for (int i = 0; i < 10; ++i) {
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (p->stop) return;
        }
        const uint64_t now = p->n_ticks.load();
        quiet = now == last ? quiet + 1 : 0;
        last = now;
        if (quiet < secs || quiet % secs) continue;
        lane_dump(p, text.data(), text.size());
        std::fprintf(stderr, "[sk_pipeline watchdog] no tick for %d s\n%s", quiet, text.data());
    }
}

void lane_destroy(sk_lane *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> a(p->rq_mu);
        std::lock_guard<std::mutex> b(p->batch_mu);
        p->stop = true;
    }
    p->rq_cv.notify_all();
    p->batch_cv.notify_all();
    p->room_cv.notify_all();
    p->deliver_cv.notify_all();
    for (std::thread &t : p->workers) t.join();
    if (p->submitter.joinable()) p->submitter.join();
    for (std::thread &t : p->deliverers) t.join();
    if (p->watchdog.joinable()) p->watchdog.join();
    for (auto &s : p->streams) release_device_side(p, *s);
    for (Batch &b : p->batches) {
        if (b.coeffs) (void)hipHostFree(b.coeffs);
        if (b.au_bytes) (void)hipHostFree(b.au_bytes);
        if (b.out_pinned) (void)hipHostFree(b.out_pinned);
        if (b.mp3_is) (void)hipHostFree(b.mp3_is);
    }
    if (p->mp3_cb) sk_mp3_codebook_destroy(p->mp3_cb);
    delete p;
}

int lane_spawn(sk_lane *p, const sk_decode_options *opt, uint32_t *handle) {
    if (!p || !handle) return SK_ERR_INVALID_ARG;
    if (const int dead = p->fatal.load()) return dead;
    sk_decode_options o{};
    if (opt) o = *opt;
    // apply_output_options' argument checks (lib.rs:3360-3376), made when the pipeline is created
    if (o.output_bits_per_sample && o.output_bits_per_sample != 16 && o.output_bits_per_sample != 24 && o.output_bits_per_sample != 32)
        return SK_ERR_UNSUPPORTED;
    uint32_t h;
    {
        std::lock_guard<std::mutex> lk(p->handles_mu);
        if (p->free_handles.empty()) return SK_ERR_CAPACITY;
        h = p->free_handles.back();
        p->free_handles.pop_back();
    }
    PStream &s = *p->streams[h];
    std::lock_guard<std::mutex> lk(s.mu);
    s.open = true;
    s.busy = s.queued = s.finished = s.cancelled = false;
    s.out_listed = false;
    s.in.clear();
    s.out.clear();
    s.opt = o;
    s.queued_bytes.store(0);
    s.pending.clear();
    s.pending_pos = 0;
    s.saw_eof = false;
    s.more = false;
    s.rate = 0;
    s.channels = 0;
    s.resample = false;
    s.codec = 0;
    s.mp3_reservoir.clear();
    s.mp3_free_format = 0;
    *handle = h;
    return SK_OK;
}

int lane_send(sk_lane *p, uint32_t handle, const uint8_t *data, size_t len) {
    PStream *s = stream_of(p, handle);
    if (!s || (len && !data)) return SK_ERR_INVALID_ARG;
    if (len > kMaxInputChunkBytes) return SK_PIPE_CHUNK_TOO_LARGE;  // lib.rs:2796-2798
    if (p->fatal.load()) return SK_PIPE_CLOSED;  // the lane's threads are gone: Disconnected
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->open || s->cancelled) return SK_PIPE_CLOSED;
    if (s->finished) return SK_PIPE_CLOSED;  // the worker has ended: TrySendError::Disconnected, lib.rs:2827-2833
    if (len && s->queued_bytes.load() + len > kMaxQueuedInputBytes) return SK_PIPE_INPUT_FULL;  // lib.rs:2800-2811
    if (s->in.size() >= p->cfg.input_buffer) return SK_PIPE_INPUT_FULL;                         // lib.rs:2820-2826
    s->in.emplace_back(data, data + len);
    s->queued_bytes.fetch_add(len);
    maybe_schedule(p, *s, handle);
    return SK_OK;
}

int take_output(sk_lane *p, PStream &s, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info) {
    // s.mu held
    if (s.out.empty()) return s.finished ? SK_PIPE_CLOSED : 0;
    Output &o = s.out.front();
    info->sampling_rate = o.rate;
    info->frames = o.frames;
    info->bytes = (uint32_t)o.data.size();
    info->bits_per_sample = o.bits;
    info->channel_count = o.channels;
    info->is_error = o.is_error ? 1 : 0;
    info->reserved = 0;
    info->status = o.status;
    if (o.data.size() > cap) return SK_ERR_CAPACITY;
    if (!o.data.empty()) std::memcpy(data, o.data.data(), o.data.size());
    s.out.pop_front();
    maybe_schedule(p, s, handle);
    if (!s.out_listed && (!s.out.empty() || s.finished)) {  // a waiter that stops early hears about the rest again
        s.out_listed = true;
        {
            std::lock_guard<std::mutex> lk(p->oq->mu);
            p->oq->ready.push_back(handle * p->n_lanes + p->lane_index);
        }
        p->oq->cv.notify_one();
    }
    return 1;
}

int lane_try_recv(sk_lane *p, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info) {
    PStream *s = stream_of(p, handle);
    if (!s || !info || (cap && !data)) return SK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->open || s->cancelled) return SK_PIPE_CLOSED;
    return take_output(p, *s, handle, data, cap, info);
}

int lane_recv(sk_lane *p, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info, uint32_t timeout_ms) {
    PStream *s = stream_of(p, handle);
    if (!s || !info || (cap && !data)) return SK_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(s->mu);
    if (!s->open || s->cancelled) return SK_PIPE_CLOSED;
    s->cv_out.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(timeout_ms),
                         [&] { return !s->out.empty() || s->finished || s->cancelled; });
    if (s->cancelled) return SK_PIPE_CLOSED;
    return take_output(p, *s, handle, data, cap, info);
}

int lane_cancel(sk_lane *p, uint32_t handle) {  // shutdown(), lib.rs:2868-2881: also what Drop does
    PStream *s = stream_of(p, handle);
    if (!s) return SK_ERR_INVALID_ARG;
    bool release_now = false;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->open || s->cancelled) return SK_PIPE_CLOSED;
        s->cancelled = true;
        s->in.clear();
        s->out.clear();
        s->queued_bytes.store(0);
        release_now = !s->busy;  // otherwise the submission thread frees it when its batch has been delivered
        if (release_now) s->open = false;
        s->cv_out.notify_all();
    }
    if (release_now) {
        release_device_side(p, *s);
        std::lock_guard<std::mutex> lk(p->handles_mu);
        p->free_handles.push_back(handle);
    }
    return SK_OK;
}

size_t lane_queued_input_bytes(sk_lane *p, uint32_t handle) {  // lib.rs:2863-2866
    PStream *s = stream_of(p, handle);
    return s ? s->queued_bytes.load() : 0;
}

int lane_get_stats(sk_lane *p, sk_pipeline_stats *out) {
    if (!p || !out) return SK_ERR_INVALID_ARG;
    out->ticks = p->n_ticks.load();
    out->frames = p->n_frames.load();
    out->outputs = p->n_outputs.load();
    out->errors = p->n_errors.load();
    out->parse_ns = p->parse_ns.load();
    out->tick_ns = p->tick_ns.load();
    out->idle_ns = p->idle_ns.load();
    out->deliver_ns = p->deliver_ns.load();
    out->entropy_threads = p->cfg.entropy_threads;
    out->lanes = 1;
    return SK_OK;
}

}  // namespace

// ---- the pipeline: lanes behind one handle space ---------------------------------------------------------------
struct sk_pipeline {
    std::vector<sk_lane *> lanes;
    std::vector<sk_engine *> owned_engines;  // lanes 1.. run on engines of their own, on the device of the caller's engine
    // one completion queue per lane: a thread in sk_pipeline_wait_outputs waits on "its" lane's queue (threads are dealt
    // out over the lanes as they first call) and sweeps the others, so consumer threads do not meet on one mutex; one
    // consumer alone still serves every lane, at worst a millisecond late
    std::vector<std::unique_ptr<OutQueue>> oqs;
    std::atomic<uint32_t> next_lane{0}, next_waiter{0};
};

namespace {
inline sk_lane *lane_of(sk_pipeline *p, uint32_t handle, uint32_t *inner) {
    if (!p || p->lanes.empty()) return nullptr;
    const uint32_t n = (uint32_t)p->lanes.size();
    *inner = handle / n;
    return p->lanes[handle % n];
}
}  // namespace

extern "C" {

int sk_pipeline_create(sk_engine *e, const sk_pipeline_config *cfg, sk_pipeline **out) try {
    sk::abi_enter();
    if (!e || !out) return SK_ERR_INVALID_ARG;
    *out = nullptr;
    sk_pipeline_config c{};
    if (cfg) c = *cfg;
    // A stream hands a tick at most `max_stream_frames_per_tick` units and is not scheduled again until that tick has been delivered, so
    // with N streams a tick carries well under N times that: the quota decides how full the ticks are, and a tick costs ~2 ms of waits
    // and launches whatever its size.  With the GPU front-end (where the host threads only frame) the default is 32 units -- two
    // resampler rounds of four chunks; the output queue of a resampling stream has room for 64.  Whole decode, 4096 streams, one
    // lane (gpurun_out/r4_ab_quota*.log): quota 16: 5.6-5.9 M access units/s; 24: 7.3-7.7 M; 32: 7.5-8.3 M; 48: 8.5-9.5 M.
    // TWO lanes (engines) when every lane can still fill a whole tick from its own streams: one lane plans, uploads and delivers
    // while the other's tick has the device (8.2 M on one lane, 11.3 M on two, quota 32).  The engines of a device take TURNS with
    // their ticks' device work (engine.cpp, g_device_turn): two ticks on the device at the same time measured no faster (9.6-10.2 M)
    // and delivered short bursts of slightly wrong samples in a third of the streams (profiles/r04_lanes_corruption.md).  With the
    // host front-end the host threads are the limit and a second lane only splits them.
    if (!c.max_streams) c.max_streams = 1024;
    const uint32_t tick_frames = c.max_frames_per_tick ? c.max_frames_per_tick : (c.gpu_entropy == 1 ? 65536u : 16384u);
    const uint32_t stream_frames = c.max_stream_frames_per_tick ? c.max_stream_frames_per_tick : (c.gpu_entropy == 1 ? 32u : 8u);
    const uint32_t streams_per_tick = std::max(1u, tick_frames / std::max(1u, stream_frames));
    uint32_t n_lanes = c.lanes ? c.lanes : ((c.gpu_entropy == 1 && c.max_streams >= 2 * streams_per_tick) ? 2u : 1u);
    if (n_lanes > 8) return SK_ERR_INVALID_ARG;
    if (n_lanes > c.max_streams) n_lanes = c.max_streams;
    if (!c.entropy_threads) {
        const unsigned cpus = usable_cpus();  // leave room for the submission + delivery threads and the callers' own
        c.entropy_threads = cpus > 5 ? std::min(cpus - 5, 64u) : 1;
    }
    sk_pipeline *p = new (std::nothrow) sk_pipeline();
    if (!p) return SK_ERR_OOM;
    sk_pipeline_config lane_cfg = c;
    lane_cfg.lanes = 1;
    lane_cfg.max_streams = (c.max_streams + n_lanes - 1) / n_lanes;
    lane_cfg.entropy_threads = std::max(1u, c.entropy_threads / n_lanes);
    int rc = SK_OK;
    for (uint32_t i = 0; i < n_lanes && rc == SK_OK; ++i) {
        sk_engine *le = e;
        if (i > 0) {
            rc = sk_engine_create(sk_engine_device(e), std::max(lane_cfg.max_streams, 16u), &le);
            if (rc != SK_OK) break;
            p->owned_engines.push_back(le);
        }
        sk_lane *lane = nullptr;
        p->oqs.emplace_back(new OutQueue());
        rc = lane_create(le, &lane_cfg, p->oqs.back().get(), i, n_lanes, &lane);
        if (rc == SK_OK) p->lanes.push_back(lane);
    }
    if (rc != SK_OK) {
        sk_pipeline_destroy(p);
        return rc;
    }
    *out = p;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_pipeline_create");
}

void sk_pipeline_destroy(sk_pipeline *p) try {
    sk::abi_enter();
    if (!p) return;
    for (auto &q : p->oqs) {
        {
            std::lock_guard<std::mutex> lk(q->mu);
            q->stop = true;
        }
        q->cv.notify_all();
    }
    if (std::getenv("SK_PIPELINE_TRACE"))
        for (sk_lane *l : p->lanes)
            std::fprintf(stderr, "[sk_pipeline] lane %u: ticks %llu frames %llu tick_ms %.1f idle_ms %.1f deliver_ms %.1f parse_ms %.1f\n", l->lane_index,
                         (unsigned long long)l->n_ticks.load(), (unsigned long long)l->n_frames.load(), l->tick_ns.load() / 1e6,
                         l->idle_ns.load() / 1e6, l->deliver_ns.load() / 1e6, l->parse_ns.load() / 1e6);
    for (sk_lane *l : p->lanes) lane_destroy(l);
    for (sk_engine *e : p->owned_engines) sk_engine_destroy(e);
    delete p;
} catch (...) {
    (void)sk::abi_caught("sk_pipeline_destroy");
}

int sk_pipeline_spawn(sk_pipeline *p, const sk_decode_options *opt, uint32_t *handle) try {
    sk::abi_enter();
    if (!p || !handle || p->lanes.empty()) return SK_ERR_INVALID_ARG;
    const uint32_t n = (uint32_t)p->lanes.size();
    const uint32_t first = p->next_lane.fetch_add(1) % n;
    int rc = SK_ERR_CAPACITY;
    for (uint32_t k = 0; k < n; ++k) {  // round robin; a full lane passes the stream on
        const uint32_t li = (first + k) % n;
        uint32_t inner = 0;
        rc = lane_spawn(p->lanes[li], opt, &inner);
        if (rc == SK_OK) {
            *handle = inner * n + li;
            return SK_OK;
        }
        if (rc != SK_ERR_CAPACITY) return rc;
    }
    return rc;
} catch (...) {
    return sk::abi_caught("sk_pipeline_spawn");
}

int sk_pipeline_send(sk_pipeline *p, uint32_t handle, const uint8_t *data, size_t len) try {
    sk::abi_enter();
    uint32_t inner;
    sk_lane *l = lane_of(p, handle, &inner);
    return l ? lane_send(l, inner, data, len) : SK_ERR_INVALID_ARG;
} catch (...) {
    return sk::abi_caught("sk_pipeline_send");
}

int sk_pipeline_finish(sk_pipeline *p, uint32_t handle) try {
    sk::abi_enter();
    return sk_pipeline_send(p, handle, nullptr, 0);
} catch (...) {
    return sk::abi_caught("sk_pipeline_finish");
}  // lib.rs:2838-2840

int sk_pipeline_try_recv(sk_pipeline *p, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info) try {
    sk::abi_enter();
    uint32_t inner;
    sk_lane *l = lane_of(p, handle, &inner);
    return l ? lane_try_recv(l, inner, data, cap, info) : SK_ERR_INVALID_ARG;
} catch (...) {
    return sk::abi_caught("sk_pipeline_try_recv");
}

int sk_pipeline_recv(sk_pipeline *p, uint32_t handle, uint8_t *data, size_t cap, sk_audio_info *info, uint32_t timeout_ms) try {
    sk::abi_enter();
    uint32_t inner;
    sk_lane *l = lane_of(p, handle, &inner);
    return l ? lane_recv(l, inner, data, cap, info, timeout_ms) : SK_ERR_INVALID_ARG;
} catch (...) {
    return sk::abi_caught("sk_pipeline_recv");
}

int sk_pipeline_cancel(sk_pipeline *p, uint32_t handle) try {
    sk::abi_enter();
    uint32_t inner;
    sk_lane *l = lane_of(p, handle, &inner);
    return l ? lane_cancel(l, inner) : SK_ERR_INVALID_ARG;
} catch (...) {
    return sk::abi_caught("sk_pipeline_cancel");
}

int sk_pipeline_wait_outputs(sk_pipeline *p, uint32_t *handles, uint32_t cap, uint32_t timeout_ms) try {
    sk::abi_enter();
    if (!p || !handles || !cap || p->oqs.empty()) return SK_ERR_INVALID_ARG;
    const uint32_t n_q = (uint32_t)p->oqs.size();
    thread_local uint32_t waiter_id = 0xffffffffu;
    if (waiter_id == 0xffffffffu) waiter_id = p->next_waiter.fetch_add(1);
    const uint32_t home = waiter_id % n_q;
    const auto deadline = std::chrono::system_clock::now() + std::chrono::milliseconds(timeout_ms);
    uint32_t n = 0;
    for (;;) {
        bool stopped = false;
        for (uint32_t k = 0; k < n_q && n < cap; ++k) {  // the home queue first, then whatever the others hold
            OutQueue &q = *p->oqs[(home + k) % n_q];
            std::unique_lock<std::mutex> lk(q.mu, std::defer_lock);
            if (k == 0) lk.lock();
            else if (!lk.try_lock()) continue;
            stopped = stopped || q.stop;
            while (n < cap && !q.ready.empty()) {
                handles[n++] = q.ready.front();
                q.ready.pop_front();
            }
        }
        if (n || stopped) break;
        const auto now = std::chrono::system_clock::now();
        if (now >= deadline) break;
        OutQueue &q = *p->oqs[home];
        std::unique_lock<std::mutex> lk(q.mu);
        // with several lanes the wait is cut into millisecond slices so that a lone consumer also sees the other lanes
        const auto until = n_q > 1 ? std::min(deadline, now + std::chrono::milliseconds(1)) : deadline;
        q.cv.wait_until(lk, until, [&] { return q.stop || !q.ready.empty(); });
    }
    for (uint32_t i = 0; i < n; ++i) {  // handed out: the next delivery (or a partial drain) lists the handle again
        uint32_t inner;
        sk_lane *l = lane_of(p, handles[i], &inner);
        PStream *s = stream_of(l, inner);
        if (!s) continue;
        std::lock_guard<std::mutex> lk(s->mu);
        s->out_listed = false;
    }
    return (int)n;
} catch (...) {
    return sk::abi_caught("sk_pipeline_wait_outputs");
}

size_t sk_pipeline_debug_dump(sk_pipeline *p, char *buf, size_t cap) try {
    sk::abi_enter();
    if (!p || !buf || !cap) return 0;
    size_t at = 0;
    buf[0] = 0;
    for (sk_lane *l : p->lanes)
        if (at + 1 < cap) at += lane_dump(l, buf + at, cap - at);
    return at;
} catch (...) {
    (void)sk::abi_caught("sk_pipeline_debug_dump");
    return 0;
}

size_t sk_pipeline_queued_input_bytes(sk_pipeline *p, uint32_t handle) try {
    sk::abi_enter();  // lib.rs:2863-2866
    uint32_t inner;
    sk_lane *l = lane_of(p, handle, &inner);
    return l ? lane_queued_input_bytes(l, inner) : 0;
} catch (...) {
    (void)sk::abi_caught("sk_pipeline_queued_input_bytes");
    return 0;
}

int sk_pipeline_get_stats(sk_pipeline *p, sk_pipeline_stats *out) try {
    sk::abi_enter();
    if (!p || !out) return SK_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    for (sk_lane *l : p->lanes) {
        sk_pipeline_stats s{};
        (void)lane_get_stats(l, &s);
        out->ticks += s.ticks;
        out->frames += s.frames;
        out->outputs += s.outputs;
        out->errors += s.errors;
        out->parse_ns += s.parse_ns;
        out->tick_ns += s.tick_ns;
        out->idle_ns += s.idle_ns;
        out->deliver_ns += s.deliver_ns;
        out->entropy_threads += s.entropy_threads;
    }
    out->lanes = (uint32_t)p->lanes.size();
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_pipeline_get_stats");
}

int sk_debug_throw_in_thread(int n, int where) {
    g_thread_throw_where.store(where, std::memory_order_relaxed);
    return g_thread_throw_after.exchange(n < 0 ? -1 : n, std::memory_order_relaxed);
}

}  // extern "C"

// adts_decoder.cpp -- the `soundkit::audio_packet::Decoder` surface for ADTS AAC-LC on one stream.
//
// Shape and contract of soundkit-aac's AacDecoder (soundkit-aac/src/lib.rs:108-266), which the streaming worker
// drives through decode_i16_with_drain (soundkit-decoder/src/lib.rs:2150-2181): every call appends its input to an
// internal buffer (<= 4 MiB per call and buffered), decodes as many whole ADTS frames as are buffered and fit in the
// caller's output, and returns the number of interleaved samples written; 0 means "needs more input" or "drained".
// Here the frames a call finds are parsed on the host (csrc/aac_frontend.cpp) and synthesised in one batched engine
// call (sk_aac_synthesize_s16: IMDCT + window + overlap-add + float_sample_to_i16 on the GPU).
#include "../../include/soundkit_amd.h"
#include "sk_abi.h"

#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {
constexpr size_t kMaxInputChunkBytes = 4u * 1024 * 1024;  // MAX_INPUT_CHUNK_BYTES, soundkit-aac lib.rs:20
constexpr size_t kMaxBufferedBytes = 4u * 1024 * 1024;    // MAX_AAC_BUFFERED_BYTES, lib.rs:22
}

struct sk_adts_decoder {
    sk_engine *engine = nullptr;
    std::vector<uint8_t> input;  // input_buffer
    size_t pos = 0;
    sk_aac_decoder *front = nullptr;
    uint8_t asc[2] = {0, 0};
    uint32_t stream = 0xffffffffu;
    uint32_t sample_rate = 0;  // None until the first frame has been decoded (lib.rs:213-215)
    uint8_t channels = 0;
    uint32_t fe_rate = 0;
    uint8_t fe_channels = 0;
    std::string last_error;
    std::vector<float> coeffs;
    std::vector<sk_aac_frame_desc> descs;
    std::vector<int32_t> status;
};

extern "C" {

int sk_adts_decoder_create(sk_engine *e, sk_adts_decoder **out) try {
    sk::abi_enter();
    if (!e || !out) return SK_ERR_INVALID_ARG;
    sk_adts_decoder *d = new (std::nothrow) sk_adts_decoder();
    if (!d) return SK_ERR_OOM;
    d->engine = e;
    *out = d;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_adts_decoder_create");
}

void sk_adts_decoder_destroy(sk_adts_decoder *d) try {
    sk::abi_enter();
    if (!d) return;
    if (d->stream != 0xffffffffu) (void)sk_stream_close(d->engine, d->stream);
    if (d->front) sk_aac_decoder_destroy(d->front);
    delete d;
} catch (...) {
    (void)sk::abi_caught("sk_adts_decoder_destroy");
}

int sk_adts_decoder_info(const sk_adts_decoder *d, uint32_t *sample_rate, uint8_t *channels) try {
    sk::abi_enter();
    if (!d) return SK_ERR_INVALID_ARG;
    if (sample_rate) *sample_rate = d->sample_rate;
    if (channels) *channels = d->channels;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_adts_decoder_info");
}

const char *sk_adts_decoder_last_error(const sk_adts_decoder *d) try {
    sk::abi_enter();
    return d ? d->last_error.c_str() : "";
} catch (...) {
    (void)sk::abi_caught("sk_adts_decoder_last_error");
    return sk::abi_message();
}

int sk_adts_decoder_decode_i16(sk_adts_decoder *d, const uint8_t *input, size_t len, int16_t *output, size_t out_cap,
                               size_t *written) try {
    sk::abi_enter();
    if (!d || !written || (len && !input) || (out_cap && !output)) return SK_ERR_INVALID_ARG;
    *written = 0;
    auto fail = [&](int rc, const std::string &msg) {
        d->last_error = msg;
        return rc;
    };
    if (len > kMaxInputChunkBytes) return fail(SK_PIPE_CHUNK_TOO_LARGE, "AAC input chunk exceeds the 4194304 byte streaming budget");
    if (len) {
        if (d->input.size() - d->pos + len > kMaxBufferedBytes)
            return fail(SK_PIPE_INPUT_FULL, "AAC decoder buffer exceeds the 4194304 byte streaming budget");
        if (d->pos > 0) {  // drain(..consumed)
            d->input.erase(d->input.begin(), d->input.begin() + (ptrdiff_t)d->pos);
            d->pos = 0;
        }
        d->input.insert(d->input.end(), input, input + len);
    }
    // frame the buffered bytes: whole frames only, as many as fit in the output
    d->descs.clear();
    size_t n_floats = 0, room = out_cap;
    // An error ends the call, but the frames framed and parsed before it in this call have been decoded as far as the
    // reference's decoder is concerned (soundkit-aac lib.rs:150-245 returns Err after decoding them: their PCM is lost, its
    // overlap state has advanced).  Run them through the synthesis before failing so that the stream's carried state
    // matches; what lands in `output` is not reported (*written stays 0).
    auto fail_after_parsed = [&](int rc, const std::string &msg) {
        if (!d->descs.empty()) {
            d->status.assign(d->descs.size(), 0);
            (void)sk_aac_synthesize_s16(d->engine, d->descs.data(), d->coeffs.data(), output, (uint32_t)d->descs.size(), d->status.data());
        }
        return fail(rc, msg);
    };
    for (;;) {
        size_t avail = d->input.size() - d->pos, frame_len = 0, pay_off = 0, pay_len = 0;
        uint8_t asc[2];
        bool have = false;
        while (avail >= 7) {
            if (sk_adts_parse(d->input.data() + d->pos, avail, &frame_len, &pay_off, &pay_len, asc) != SK_OK) {
                d->pos += 1;  // resynchronise on the next syncword
                avail -= 1;
                continue;
            }
            have = avail >= frame_len;
            break;
        }
        if (!have) break;  // NOT_ENOUGH_BITS: needs more data
        if (!d->front) {
            int rc = sk_aac_decoder_create(asc, 2, &d->front);
            if (rc != SK_OK) return fail(rc, std::string("Decoding error: ") + sk_strerror(rc));
            d->asc[0] = asc[0];
            d->asc[1] = asc[1];
            (void)sk_aac_decoder_info(d->front, &d->fe_rate, &d->fe_channels);
            rc = sk_stream_open(d->engine, d->fe_rate, d->fe_channels, &d->stream);
            if (rc != SK_OK) return fail(rc, std::string("Decoding error: ") + sk_strerror(rc));
        } else if (asc[0] != d->asc[0] || asc[1] != d->asc[1]) {
            return fail_after_parsed(SK_AAC_ERR_UNSUPPORTED_FEATURE, "Decoding error: AAC configuration changed mid-stream");
        }
        const size_t frame_samples = (size_t)d->fe_channels * SK_AAC_FRAME_LEN;
        if (room == 0) break;
        if (room < frame_samples)
            return fail_after_parsed(SK_ERR_CAPACITY, "Output buffer too small for decoded frame (needed " + std::to_string(frame_samples) +
                                                          ", had " + std::to_string(room) + ")");
        if (d->coeffs.size() < n_floats + frame_samples) d->coeffs.resize(n_floats + frame_samples + 16 * frame_samples);
        sk_aac_frame_desc desc{};
        const int rc = sk_aac_decoder_parse(d->front, d->input.data() + d->pos + pay_off, pay_len, d->coeffs.data() + n_floats, &desc);
        if (rc != SK_OK) {
            d->pos += frame_len;  // the frame is consumed either way
            return fail_after_parsed(rc, std::string("Decoding error: ") + sk_aac_decoder_last_error(d->front));
        }
        desc.stream = d->stream;
        d->descs.push_back(desc);
        n_floats += frame_samples;
        room -= frame_samples;
        d->pos += frame_len;
    }
    if (d->descs.empty()) return SK_OK;
    d->status.assign(d->descs.size(), 0);
    const int rc = sk_aac_synthesize_s16(d->engine, d->descs.data(), d->coeffs.data(), output, (uint32_t)d->descs.size(), d->status.data());
    if (rc != SK_OK) return fail(rc, std::string("Decoding error: ") + sk_strerror(rc));
    for (int32_t st : d->status)
        if (st != 0) return fail(SK_ERR_INVALID_ARG, "Decoding error: frame rejected by the synthesis engine");
    d->sample_rate = d->fe_rate;
    d->channels = d->fe_channels;
    *written = n_floats;  // one i16 per spectral coefficient: channels * 1024 per frame
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_adts_decoder_decode_i16");
}

int sk_adts_decoder_decode_f32(sk_adts_decoder *d, const uint8_t *input, size_t len, float *output, size_t out_cap, size_t *written) try {
    sk::abi_enter();
    if (!d || !written || (out_cap && !output)) return SK_ERR_INVALID_ARG;
    std::vector<int16_t> tmp(out_cap);  // lib.rs:255-265: decode to i16, then / 32768
    const int rc = sk_adts_decoder_decode_i16(d, input, len, tmp.data(), out_cap, written);
    if (rc != SK_OK) return rc;
    for (size_t i = 0; i < *written; ++i) output[i] = (float)tmp[i] / 32768.0f;
    return SK_OK;
} catch (...) {
    return sk::abi_caught("sk_adts_decoder_decode_f32");
}

}  // extern "C"

// Mutation fuzz of the host side of the MP3 path -- frame scan, header, side information, bit reservoir, scale factors,
// the Huffman stage over a code book, and the decoder handle's buffering -- built with AddressSanitizer + UBSan on the CPU
// (tests/test_mp3_decoder.py::test_mutated_streams_under_sanitizers).  Mp3Decoder is fed bytes from the network
// (soundkit-mp3/src/lib.rs:279-305): whatever arrives, every call must come back with samples or a status, with no
// out-of-bounds access and no undefined behaviour.  The GPU stages are stubs here (they get checked shapes and return
// silence); two code books take turns: the standard's (csrc/mp3_iso_tables.h -- the reference's files decode through the
// whole Huffman stage with it) and a synthetic one (fixed-length codes: complete for the tables of 4, 16, 64 and 256
// symbols, not for those of 9 and 36, so that both "decoded" and "no such code" are reached on the streams written with it).
// The stub stages also (a) fail now and then -- the call must leave the decoder as it was: repeating it gives what a clean
// decoder gives; (b) reject single granules -- that frame's samples must be missing from the output, nothing else;
// (c) check that every granule names a stream that was opened with the granule's channel count (a mono -> stereo splice
// must not queue granules of both kinds in one launch).
//   usage: fuzz_mp3 ITERATIONS file.mp3...
#include "../soundkit_amd/csrc/mp3_bitstream.cpp"
#include "../soundkit_amd/csrc/mp3_decoder.cpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

// a failed check says where
#define abort() (std::fprintf(stderr, "fuzz_mp3: check failed at line %d\n", __LINE__), std::abort())

// ---- what mp3_decoder.cpp calls in the engine ---------------------------------------------------------------------------------
struct sk_engine {
    int open = 0;
    uint32_t next_id = 0;
    std::map<uint32_t, uint8_t> channels;
};
static sk_engine *g_engine = nullptr;
static size_t g_granules = 0, g_calls = 0, g_failed_calls = 0, g_rejected = 0;
static int g_fail_in = -1;    // >= 0: the g_fail_in-th stage call from now fails as a whole
static int g_reject_in = -1;  // >= 0: the g_reject_in-th granule from now comes back with a status
extern "C" {
int sk_stream_open(sk_engine *e, uint32_t, uint8_t channels, uint32_t *out) {
    if (channels < 1 || channels > 2) abort();
    *out = e->next_id++;
    e->channels[*out] = channels;
    ++e->open;
    return SK_OK;
}
int sk_stream_close(sk_engine *e, uint32_t id) {
    if (!e->channels.erase(id)) abort();
    --e->open;
    return SK_OK;
}
int sk_mp3_set_band_tables(sk_engine *, uint32_t, const uint16_t *, const uint16_t *, const uint8_t *) { return SK_OK; }
int sk_mp3_set_synthesis_window(sk_engine *, const float *) { return SK_OK; }
static int stage(const sk_mp3_requant_granule *g, const sk_mp3_granule_desc *d, const int16_t *is, void *pcm, uint32_t n, int32_t *status, size_t width) {
    size_t rows = 0;
    if (g_fail_in == 0) {
        g_fail_in = -1;
        ++g_failed_calls;
        return SK_ERR_HIP;
    }
    if (g_fail_in > 0) --g_fail_in;
    for (uint32_t i = 0; i < n; ++i) {
        if (g[i].channels < 1 || g[i].channels > 2 || d[i].channels != g[i].channels) abort();
        const auto opened = g_engine->channels.find(d[i].stream);
        if (opened == g_engine->channels.end() || opened->second != d[i].channels) abort();  // a granule for a stream of another shape
        for (int c = 0; c < g[i].channels; ++c) {
            if (g[i].ch[c].block_type > 3 || d[i].block_type[c] != g[i].ch[c].block_type) abort();
            for (int k = 0; k < 576; ++k) {
                const int v = is[(rows + c) * 576 + k];
                if (v > 8206 || v < -8206) abort();  // 15 + the widest escape: nothing larger can come out of the Huffman stage
            }
        }
        rows += g[i].channels;
        status[i] = 0;
        if (g_reject_in == 0) status[i] = SK_MP3_UNSUPPORTED, ++g_rejected;
        if (g_reject_in >= 0) --g_reject_in;
    }
    // samples that say which granule they belong to, so that a frame taken out of the output can be told from its neighbours
    size_t row = 0;
    for (uint32_t i = 0; i < n; ++i)
        for (size_t k = 0; k < 576u * g[i].channels; ++k, ++row) {
            const int v = status[i] ? 0x7fff : (int)(i & 0xff) + 1;
            if (width == 2) ((int16_t *)pcm)[row] = (int16_t)v;
            else ((float *)pcm)[row] = (float)v * (1.0f / 32768.0f);
        }
    g_granules += n;
    ++g_calls;
    return SK_OK;
}
int sk_mp3_decode_granules_f32(sk_engine *, const sk_mp3_requant_granule *g, const sk_mp3_granule_desc *d, const int16_t *is, float *pcm, uint32_t n,
                               int32_t *status) {
    return stage(g, d, is, pcm, n, status, 4);
}
int sk_mp3_decode_granules_s16(sk_engine *, const sk_mp3_requant_granule *g, const sk_mp3_granule_desc *d, const int16_t *is, int16_t *pcm, uint32_t n,
                               int32_t *status) {
    return stage(g, d, is, pcm, n, status, 2);
}
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t next() {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    return (uint32_t)(rng >> 16);
}

int main(int argc, char **argv) {
    const int iters = atoi(argv[1]);
    // the code book: fixed-length codes (value = symbol), so 2^len > symbols leaves bit patterns that are no code
    static const uint8_t xlen[32] = {0, 2, 3, 3, 0, 4, 4, 6, 6, 6, 8, 8, 8, 16, 0, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    static const uint8_t linbits[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13};
    std::vector<std::vector<uint8_t>> hlen(32);
    std::vector<std::vector<uint32_t>> hcod(32);
    sk_mp3_tables t;
    std::memset(&t, 0, sizeof t);
    for (int i = 0; i < 32; ++i) {
        if (!xlen[i]) continue;
        const int n = xlen[i] * xlen[i];
        int len = 1;
        while ((1 << len) < n) ++len;  // tables of 9 and 36 symbols leave bit patterns that are no code
        hlen[i].assign(n, (uint8_t)len);
        hcod[i].resize(n);
        for (int s = 0; s < n; ++s) hcod[i][s] = (uint32_t)s;
        t.big_values[i] = sk_mp3_code_table{xlen[i], linbits[i], hlen[i].data(), hcod[i].data()};
    }
    for (int k = 0; k < 2; ++k)
        for (int s = 0; s < 16; ++s) t.count1_hlen[k][s] = (uint8_t)(k ? 4 : 5), t.count1_hcod[k][s] = (uint8_t)s;
    static const uint8_t slen[16][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3}, {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
    std::memcpy(t.slen, slen, sizeof slen);
    static const uint8_t parts[6][3][4] = {{{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},     {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
                                           {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
                                           {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}},     {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};
    std::memcpy(t.lsf_partitions, parts, sizeof parts);
    for (int row = 0; row < 9; ++row) {
        t.rates_present[row] = row != 4;  // one rate without tables: SK_MP3_UNSUPPORTED on that path
        for (int i = 0; i < 22; ++i) t.long_offsets[row][i] = (uint16_t)(i < 9 ? 4 * i + (i == 8 ? 4 : 0) : 36 + 41 * (i - 8));
        t.long_offsets[row][8] = 36, t.long_offsets[row][22] = 576;
        for (int i = 0; i < 13; ++i) t.short_offsets[row][i] = (uint16_t)(i < 4 ? 4 * i : 12 + 15 * (i - 3));
        t.short_offsets[row][13] = 192;
    }
    sk_mp3_codebook *cb = nullptr;
    if (sk_mp3_codebook_create(&t, &cb) != SK_OK) {
        std::fprintf(stderr, "codebook rejected\n");
        return 2;
    }
    sk_mp3_codebook *iso = nullptr;
    if (sk_mp3_codebook_create_iso(&iso) != SK_OK) {
        std::fprintf(stderr, "the standard's tables rejected\n");
        return 2;
    }
    sk_engine engine;
    g_engine = &engine;
    size_t calls = 0, samples = 0, errors = 0, retried = 0, spliced = 0;
    std::vector<int16_t> out16(1 << 15);
    std::vector<float> out32(1 << 15);
    std::vector<int32_t> outi(1 << 15);
    for (int a = 2; a < argc; ++a) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) return 2;
        std::vector<uint8_t> clean(1 << 20);
        clean.resize(std::fread(clean.data(), 1, clean.size(), f));
        std::fclose(f);
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> d = clean;
            const uint32_t kind = it == 0 ? 99 : next() % 6;
            const uint32_t hits = 1 + next() % 24;
            for (uint32_t h = 0; h < hits && kind != 99; ++h) {
                const size_t at = next() % d.size();
                if (kind == 0) d[at] ^= (uint8_t)(1u << (next() & 7));
                else if (kind == 1) d[at] = (uint8_t)next();
                else if (kind == 2) d.erase(d.begin() + (ptrdiff_t)at, d.begin() + (ptrdiff_t)std::min(d.size(), at + 1 + next() % 40));
                else if (kind == 3) d.insert(d.begin() + (ptrdiff_t)at, (size_t)(1 + next() % 40), (uint8_t)next());
                else if (kind == 4 && at + 4 < d.size()) {  // headers of other versions / modes / rates in front of real side information
                    d[at] = 0xff;
                    d[at + 1] = (uint8_t)(0xe0 | (next() & 0x1f));
                    d[at + 2] = (uint8_t)next();
                    d[at + 3] = (uint8_t)next();
                } else if (kind == 5) d.resize(1 + next() % d.size());
                if (d.empty()) d.push_back(0);
            }
            if (it % 5 == 3 && a + 1 < argc) {  // a splice: another file's frames behind this one's (mono -> stereo among them)
                FILE *g = std::fopen(argv[a + 1], "rb");
                if (!g) return 2;
                std::vector<uint8_t> more(1 << 20);
                more.resize(std::fread(more.data(), 1, more.size(), g));
                std::fclose(g);
                d.resize(std::min<size_t>(d.size(), 1 + next() % d.size()));
                d.insert(d.end(), more.begin(), more.end());
                ++spliced;
            }
            sk_mp3_decoder *dec = nullptr;
            if (sk_mp3_decoder_create(&engine, (it & 1) ? iso : cb, &dec) != SK_OK) return 2;
            if (it % 4 == 2) {
                // a stage call that fails must leave the decoder as it was: the same call again gives what a decoder that never
                // failed gives -- same sample count, same samples (the stub's samples carry their granule's place in the launch)
                sk_mp3_decoder *twin = nullptr;
                if (sk_mp3_decoder_create(&engine, (it & 1) ? iso : cb, &twin) != SK_OK) return 2;
                std::vector<int16_t> a16(1 << 15), b16(1 << 15);
                size_t at = 0;
                while (at < d.size()) {
                    const size_t n = std::min<size_t>(d.size() - at, 1 + next() % 3000);
                    size_t wa = 0, wb = 0;
                    g_fail_in = (int)(next() % 2);  // this call's launch, or (if it makes none) a later one
                    int rc = sk_mp3_decoder_decode_i16(dec, d.data() + at, n, a16.data(), a16.size(), &wa);
                    if (rc == SK_ERR_HIP) {
                        if (wa) abort();
                        g_fail_in = -1;
                        rc = sk_mp3_decoder_decode_i16(dec, d.data() + at, n, a16.data(), a16.size(), &wa);
                        ++retried;
                    }
                    g_fail_in = -1;
                    const int rc2 = sk_mp3_decoder_decode_i16(twin, d.data() + at, n, b16.data(), b16.size(), &wb);
                    if (rc != rc2 || wa != wb || std::memcmp(a16.data(), b16.data(), wa * 2)) abort();
                    size_t ba = 0, bb = 0;
                    uint64_t fa = 0, fb = 0;
                    sk_mp3_decoder_info(dec, nullptr, nullptr, &ba, &fa);
                    sk_mp3_decoder_info(twin, nullptr, nullptr, &bb, &fb);
                    if (ba != bb || fa != fb) abort();
                    at += n;
                }
                sk_mp3_decoder_destroy(twin);
                sk_mp3_decoder_reset(dec);
            }
            if (it % 6 == 1) g_reject_in = (int)(next() % 40);  // one granule of this run comes back rejected
            size_t pos = 0;
            const int which = (int)(next() % 3);
            while (pos < d.size()) {
                const size_t n = std::min<size_t>(d.size() - pos, 1 + next() % 3000);
                const size_t cap = next() % 8 == 0 ? 1 + next() % 3000 : out16.size();  // sometimes a buffer no frame fits in
                size_t written = 0;
                int rc;
                if (which == 0) rc = sk_mp3_decoder_decode_i16(dec, d.data() + pos, n, out16.data(), cap, &written);
                else if (which == 1) rc = sk_mp3_decoder_decode_f32(dec, d.data() + pos, n, out32.data(), cap, &written);
                else rc = sk_mp3_decoder_decode_i32(dec, d.data() + pos, n, outi.data(), cap, &written);
                if (written > cap) abort();
                if (rc != SK_OK && rc != SK_ERR_CAPACITY) abort();  // nothing else can come out of this path with the stub stages
                for (size_t k = 0; k < written && which == 0; ++k)
                    if (out16[k] == 0x7fff) abort();  // samples of a rejected granule were handed out
                errors += rc != SK_OK;
                samples += written;
                pos += n;
                ++calls;
            }
            for (int drain = 0; drain < 64; ++drain) {
                size_t written = 0;
                const int rc = sk_mp3_decoder_decode_i16(dec, nullptr, 0, out16.data(), out16.size(), &written);
                if (rc != SK_OK) abort();
                samples += written;
                if (!written) break;
            }
            size_t buffered = 0;
            sk_mp3_decoder_info(dec, nullptr, nullptr, &buffered, nullptr);
            // at most one incomplete frame stays behind -- or what a free-format header waits for: the next two headers of its stream,
            // each up to 2304 bytes on (sk_mp3_scan_free)
            if (buffered > 2 * 2304 + 8) abort();
            g_reject_in = -1;
            if (it % 7 == 0) sk_mp3_decoder_reset(dec);
            sk_mp3_decoder_destroy(dec);
            if (engine.open != 0) abort();
        }
    }
    sk_mp3_codebook_destroy(cb);
    sk_mp3_codebook_destroy(iso);
    std::printf("calls %zu samples %zu capacity-errors %zu granules %zu gpu-calls %zu failed-calls %zu retried %zu rejected-granules %zu splices %zu\n", calls, samples,
                errors, g_granules, g_calls, g_failed_calls, retried, g_rejected, spliced);
    return 0;
}

// How fast is the host half of the MP3 path (frame scan, side information, reservoir, scale factors, Huffman stage)?
//   g++ -O2 -std=c++17 -o /tmp/mp3_host_rate tools/mp3_host_rate.cpp && /tmp/mp3_host_rate tests/golden/mp3/stereo16k_A_Tusk_encoded.mp3
// The two product sources are compiled in; the engine entry points they reference are stubs that are never called here.
#include "../soundkit_amd/csrc/mp3_bitstream.cpp"
#include "../soundkit_amd/csrc/mp3_decoder.cpp"

#include <chrono>
#include <cstdio>

struct sk_engine {};
extern "C" {
int sk_stream_open(sk_engine *, uint32_t, uint8_t, uint32_t *) { return SK_ERR_UNSUPPORTED; }
int sk_stream_close(sk_engine *, uint32_t) { return SK_OK; }
int sk_mp3_set_band_tables(sk_engine *, uint32_t, const uint16_t *, const uint16_t *, const uint8_t *) { return SK_OK; }
int sk_mp3_set_synthesis_window(sk_engine *, const float *) { return SK_OK; }
int sk_mp3_decode_granules_f32(sk_engine *, const sk_mp3_requant_granule *, const sk_mp3_granule_desc *, const int16_t *, float *, uint32_t, int32_t *) { return SK_ERR_UNSUPPORTED; }
int sk_mp3_decode_granules_s16(sk_engine *, const sk_mp3_requant_granule *, const sk_mp3_granule_desc *, const int16_t *, int16_t *, uint32_t, int32_t *) { return SK_ERR_UNSUPPORTED; }
}

int main(int argc, char **argv) {
    if (argc < 2) return 64;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 65;
    std::vector<uint8_t> data(1 << 22);
    data.resize(std::fread(data.data(), 1, data.size(), f));
    std::fclose(f);
    sk_mp3_codebook *cb = nullptr;
    if (sk_mp3_codebook_create_iso(&cb) != SK_OK) return 2;
    std::vector<sk_mp3_frame_info> frames(4096);
    uint32_t n = 0;
    size_t used = 0;
    if (sk_mp3_scan(data.data(), data.size(), frames.data(), (uint32_t)frames.size(), &n, &used) != SK_OK) return 3;
    const int loops = 400;
    size_t granules = 0, bad = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int loop = 0; loop < loops; ++loop) {
        std::vector<uint8_t> reservoir, main(8192);
        for (uint32_t k = 0; k < n; ++k) {
            const sk_mp3_frame_info &h = frames[k];
            const uint8_t *frame = data.data() + h.offset;
            sk_mp3_side_info side;
            if (sk_mp3_parse_side_info(frame, h.frame_bytes, &h, &side) != SK_OK) { ++bad; continue; }
            size_t main_len = 0;
            main.resize(reservoir.size() + h.frame_bytes);
            const bool ok = sk_mp3_main_data(frame, h.frame_bytes, &h, &side, reservoir.data(), reservoir.size(), main.data(), main.size(), &main_len) == SK_OK;
            sk_mp3_granule_data out[2][2];
            if (ok && sk_mp3_decode_main_data(cb, &h, &side, main.data(), main_len, out) == SK_OK) granules += (size_t)h.granules * h.channels;
            else ++bad;
            const size_t head = 4u + (h.has_crc ? 2u : 0u) + h.side_info_bytes;
            reservoir.insert(reservoir.end(), frame + head, frame + h.frame_bytes);
            if (reservoir.size() > 2048) reservoir.erase(reservoir.begin(), reservoir.end() - 2048);
        }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%u frames x %d loops: %.3f s, %.2f us per granule-channel, %.2f us per frame (%zu undecodable)\n", n, loops, dt,
                dt * 1e6 / (double)granules, dt * 1e6 / ((double)n * loops), bad);
    sk_mp3_codebook_destroy(cb);
    return 0;
}

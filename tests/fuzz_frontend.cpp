// Mutation fuzz of the AAC-LC access-unit front-end, built with AddressSanitizer + UBSan on the CPU
// (tests/test_aac_frontend.py::test_mutated_access_units_under_sanitizers).  The reference promises that
// malformed input never panics (soundkit-aac-lc/tests/malformed_decode.rs); here that means: every mutated access
// unit is either decoded or rejected with an error code, with no out-of-bounds access and no undefined behaviour.
//   usage: fuzz_frontend ITERATIONS_PER_FILE file.adts...
#include "../soundkit_amd/csrc/aac_frontend.cpp"
#include <cstdio>
#include <vector>
static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }
int main(int argc, char **argv) {
    int iters = atoi(argv[1]);
    size_t ok = 0, err = 0;
    for (int a = 2; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        std::vector<uint8_t> d(1 << 20);
        size_t n = fread(d.data(), 1, d.size(), f);
        fclose(f);
        std::vector<std::vector<uint8_t>> aus;
        uint8_t asc[2];
        size_t pos = 0;
        while (pos + 7 <= n) {
            size_t fl, po, pl;
            if (sk_adts_parse(d.data() + pos, n - pos, &fl, &po, &pl, asc) != 0 || pos + fl > n) break;
            aus.emplace_back(d.begin() + pos + po, d.begin() + pos + po + pl);
            pos += fl;
        }
        sk_aac_decoder *dec = nullptr;
        sk_aac_decoder_create(asc, 2, &dec);
        std::vector<float> coeffs(2048);
        sk_aac_frame_desc desc;
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> au = aus[next() % aus.size()];
            int flips = 1 + next() % 6;
            for (int k = 0; k < flips; ++k) {
                uint32_t r = next();
                if (au.empty()) break;
                switch (r % 4) {
                case 0: au[(r >> 8) % au.size()] ^= (uint8_t)(1u << ((r >> 4) & 7)); break;
                case 1: au[(r >> 8) % au.size()] = (uint8_t)(r >> 20); break;
                case 2: au.resize((r >> 8) % (au.size() + 1)); break;
                default: { size_t p = (r >> 8) % au.size(); au.insert(au.begin() + p, (uint8_t)(r >> 20)); } break;
                }
            }
            int rc = sk_aac_decoder_parse(dec, au.data(), au.size(), coeffs.data(), &desc);
            if (rc == 0) {
                ++ok;
                for (float v : coeffs) if (!(v == v)) { /* NaN allowed? count */ }
                if (desc.window_sequence[0] > 3 || desc.window_sequence[1] > 3 || desc.window_shape[0] > 1 || desc.window_shape[1] > 1) { printf("bad desc\n"); return 2; }
            } else ++err;
        }
        sk_aac_decoder_destroy(dec);
    }
    printf("ok %zu err %zu\n", ok, err);
}

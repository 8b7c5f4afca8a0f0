// Zero heap allocation on the decode path after warm-up, as soundkit-aac-lc/tests/no_alloc_decode.rs requires of
// the reference's AacLcDecoder: the front-end (csrc/aac_frontend.cpp) parses every access unit of the fixtures once to
// warm up, then again with every operator new / malloc counted.   usage: noalloc_frontend file.adts...
#include "../soundkit_amd/csrc/aac_frontend.cpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

static std::atomic<bool> g_counting{false};
static std::atomic<size_t> g_allocs{0};
void *operator new(size_t n) {
    if (g_counting.load()) g_allocs.fetch_add(1);
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete(void *p, size_t) noexcept { std::free(p); }

int main(int argc, char **argv) {
    size_t frames = 0;
    for (int a = 1; a < argc; ++a) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) return 2;
        std::vector<uint8_t> d(1 << 20);
        const size_t n = std::fread(d.data(), 1, d.size(), f);
        std::fclose(f);
        sk_aac_decoder *dec = nullptr;
        std::vector<float> coeffs(2048);
        sk_aac_frame_desc desc;
        for (int pass = 0; pass < 2; ++pass) {
            g_counting.store(pass == 1);
            size_t pos = 0;
            while (pos + 7 <= n) {
                size_t fl, po, pl;
                uint8_t asc[2];
                if (sk_adts_parse(d.data() + pos, n - pos, &fl, &po, &pl, asc) != 0 || pos + fl > n) break;
                if (!dec && sk_aac_decoder_create(asc, 2, &dec) != 0) return 3;
                if (sk_aac_decoder_parse(dec, d.data() + pos + po, pl, coeffs.data(), &desc) != 0) return 4;
                pos += fl;
                frames += pass;
            }
            g_counting.store(false);
        }
        sk_aac_decoder_destroy(dec);
    }
    std::printf("%zu frames, %zu allocations after warm-up\n", frames, g_allocs.load());
    return g_allocs.load() == 0 ? 0 : 1;
}

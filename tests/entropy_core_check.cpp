// CPU equivalence check of the host+device entropy core (csrc/aac_entropy_core.h) against the product front-end
// (csrc/aac_frontend.cpp): same status code, bit-identical spectra and the same window fields on every access unit of
// the fixtures and on mutated access units.  Built with AddressSanitizer + UBSan by tests/test_entropy_core.py.
//   usage: entropy_core_check MUTANTS_PER_FILE file.adts...
#include "../soundkit_amd/csrc/aac_frontend.cpp"
#include "../soundkit_amd/csrc/aac_entropy_core.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static size_t g_wide_values = 0;  // quantised values beyond i16 that went through the hand-over's list
static uint32_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }

static sk_ec::Tables make_tables() {
    const sk_ec::HostTables &h = sk_ec::host_tables();
    for (int b = 0; b < 12; ++b)
        if (h.primary_bits[b] != sk_ec::kPrimaryBits) abort();
    sk_ec::Tables t{};
    t.meta = h.meta.data();
    t.lut = h.lut.data();
    t.tuples = h.tuples.data();
    t.swb = h.swb.data();
    t.pow43 = h.pow43.data();
    t.pow43_lo = h.pow43.data();
    t.sf_mult = h.sf_mult.data();
    t.is_mult = h.is_mult.data();
    t.tns_sin = h.tns_sin.data();
    t.sf_wide = h.sf_wide.data();
    t.is_wide = h.is_wide.data();
    return t;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int mutants = atoi(argv[1]);
    const sk_ec::Tables tables = make_tables();
    size_t checked = 0, accepted = 0;
    for (int a = 2; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 2;
        std::vector<uint8_t> d(1 << 20);
        const size_t n = fread(d.data(), 1, d.size(), f);
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
        if (sk_aac_decoder_create(asc, 2, &dec) != 0) return 3;
        sk_ec::Stream st{dec->d.sf_index, dec->d.channels, 0x1f2e3d4cu};
        std::vector<float> want(2048), got(2048);
        std::vector<uint32_t> words;
        sk_ec::Scratch scratch;
        auto compare = [&](const std::vector<uint8_t> &au, const char *what, size_t index) -> bool {
            // both decoders carry PNS state across access units: keep them in step
            dec->d.pns_state = st.pns_state;
            sk_aac_frame_desc desc{};
            std::fill(want.begin(), want.end(), 0.0f);
            std::fill(got.begin(), got.end(), 0.0f);
            const int rc_want = sk_aac_decoder_parse(dec, au.data(), au.size(), want.data(), &desc);
            words.assign((au.size() + 3) / 4 + 2, 0);  // 4-byte aligned, >= 8 bytes of zero padding
            memcpy(words.data(), au.data(), au.size());
            uint8_t seq[2] = {0, 0}, shape[2] = {0, 0};
            const uint32_t pns_before = st.pns_state;
            const int rc_got = sk_ec::decode_access_unit(tables, st, words.data(), (uint32_t)au.size(), got.data(), seq, shape, scratch);
            ++checked;
            {  // the frame-parallel composition: count the noise, jump the generator, fill and finish -- must equal the above
                sk_ec::Stream st2{st.sf_index, st.channels, pns_before};
                std::vector<float> split(2048, 0.0f);
                uint8_t seq2[2] = {0, 0}, shape2[2] = {0, 0};
                sk_ec::Scratch scratch2;
                int rc2 = sk_ec::parse_unit(tables, st2, words.data(), (uint32_t)au.size(), split.data(), seq2, shape2, scratch2, sk_ec::PNS_COUNT);
                if (rc2 == 0 && st2.pns_state != pns_before) { printf("%s %zu: counting phase moved the generator\n", what, index); return false; }
                if (rc2 == 0) rc2 = sk_ec::finish_unit(tables, st2, words.data(), (uint32_t)au.size(), split.data(), scratch2, true);
                if (rc2 != rc_got) { printf("%s %zu: split status %d vs %d\n", what, index, rc2, rc_got); return false; }
                if (rc2 == 0) {
                    if (memcmp(split.data(), got.data(), sizeof(float) * 1024 * (size_t)st.channels) != 0 || memcmp(seq, seq2, 2) || memcmp(shape, shape2, 2)) {
                        printf("%s %zu: split decode differs\n", what, index);
                        return false;
                    }
                    if (st2.pns_state != st.pns_state || sk_ec::pns_advance(pns_before, scratch2.noise_samples) != st.pns_state) {
                        printf("%s %zu: generator state after the split decode / jump-ahead differs\n", what, index);
                        return false;
                    }
                }
            }
            {  // the quantised hand-over: host parse to integers -> wire record -> unpack, dequantise, finish == the above
                sk_ec::Stream st3{st.sf_index, st.channels, pns_before};
                std::vector<int16_t> quant(2048, 0), sf0(128, 0), sf1(128, 0);
                sk_ec::WideList wide_list;
                wide_list.n = 0;
                const sk_ec::QuantCapture qc{quant.data(), {sf0.data(), sf1.data()}, &wide_list};
                uint8_t seq3[2] = {0, 0}, shape3[2] = {0, 0};
                sk_ec::Scratch sc3;
                int rc3 = sk_ec::parse_unit(tables, st3, words.data(), (uint32_t)au.size(), nullptr, seq3, shape3, sc3, sk_ec::PNS_COUNT, &qc);
                // more values beyond i16 than the record's list holds (kWideMax): the one case this mode refuses on its own
                const bool wide = rc3 == sk_ec::EC_UNSUPPORTED_FEATURE && rc_got != sk_ec::EC_UNSUPPORTED_FEATURE && wide_list.n == (uint32_t)sk_ec::kWideMax;
                g_wide_values += wide_list.n;
                if (rc3 == 0) {
                    const int32_t tail = sk_ec::unit_tail(words.data(), (uint32_t)au.size(), sc3.resume_pos);
                    const int16_t *sfs[2] = {sf0.data(), sf1.data()};
                    sk_ec::WireUnit wire;
                    sk_ec::pack_unit(sc3, st.channels, sfs, tail, wire, &wide_list);
                    sk_ec::Scratch back{};
                    rc3 = sk_ec::unpack_unit(tables, wire, back);
                    if (rc3 != 0) { printf("%s %zu: the record of a parsed unit does not unpack (%d)\n", what, index, rc3); return false; }
                    std::vector<float> qout(2048, 0.0f);
                    rc3 = sk_ec::dequant_channel(tables, st3, back.ch[0], quant.data(), qout.data());
                    if (rc3 == 0 && st.channels == 2) rc3 = sk_ec::dequant_channel(tables, st3, back.ch[1], quant.data() + 1024, qout.data() + 1024);
                    if (rc3 == 0) rc3 = sk_ec::apply_wide(wire, (uint32_t)st.channels, back, tables, st3, quant.data(), qout.data());
                    static const uint32_t no_bits[4] = {0, 0, 0, 0};
                    if (rc3 == 0) rc3 = sk_ec::finish_unit(tables, st3, no_bits, 0, qout.data(), back, true);
                    if (rc3 == 0) rc3 = wire.tail_status;
                    if (rc3 != rc_got) { printf("%s %zu: quantised hand-over status %d vs %d\n", what, index, rc3, rc_got); return false; }
                    if (rc3 == 0 && (memcmp(qout.data(), got.data(), sizeof(float) * 1024 * (size_t)st.channels) != 0 || memcmp(seq, seq3, 2) ||
                                     memcmp(shape, shape3, 2) || st3.pns_state != st.pns_state)) {
                        printf("%s %zu: quantised hand-over decode differs\n", what, index);
                        return false;
                    }
                } else if (!wide && rc3 != rc_got) {
                    printf("%s %zu: quantised parse status %d vs %d\n", what, index, rc3, rc_got);
                    return false;
                }
            }
            if (rc_want != rc_got) {
                printf("%s %zu: status %d vs %d (%s)\n", what, index, rc_want, rc_got, sk_aac_decoder_last_error(dec));
                return false;
            }
            if (rc_want != 0) {
                st.pns_state = 0x1f2e3d4cu;  // a failed access unit leaves the generators wherever they stopped: restart both
                return true;
            }
            ++accepted;
            if (memcmp(want.data(), got.data(), sizeof(float) * 1024 * (size_t)st.channels) != 0) {
                for (int i = 0; i < 1024 * st.channels; ++i)
                    if (memcmp(&want[i], &got[i], 4) != 0) {
                        printf("%s %zu: coefficient %d differs: %a vs %a\n", what, index, i, want[i], got[i]);
                        break;
                    }
                return false;
            }
            for (int c = 0; c < st.channels; ++c)
                if (desc.window_sequence[c] != seq[c] || desc.window_shape[c] != shape[c]) {
                    printf("%s %zu: window fields differ\n", what, index);
                    return false;
                }
            if (dec->d.pns_state != st.pns_state) {
                printf("%s %zu: PNS state differs\n", what, index);
                return false;
            }
            return true;
        };
        for (size_t i = 0; i < aus.size(); ++i)
            if (!compare(aus[i], "frame", i)) return 1;
        for (int it = 0; it < mutants; ++it) {
            std::vector<uint8_t> au = aus[next() % aus.size()];
            const int flips = 1 + next() % 5;
            for (int k = 0; k < flips; ++k) {
                const uint32_t r = next();
                if (au.empty()) break;
                switch (r % 4) {
                case 0: au[(r >> 8) % au.size()] ^= (uint8_t)(1u << ((r >> 4) & 7)); break;
                case 1: au[(r >> 8) % au.size()] = (uint8_t)(r >> 20); break;
                case 2: au.resize((r >> 8) % (au.size() + 1)); break;
                default: { const size_t p = (r >> 8) % au.size(); au.insert(au.begin() + (ptrdiff_t)p, (uint8_t)(r >> 20)); } break;
                }
            }
            if (!compare(au, "mutant", (size_t)it)) return 1;
        }
        sk_aac_decoder_destroy(dec);
    }
    printf("checked %zu access units, %zu accepted: identical\n", checked, accepted);
    printf("wide values %zu\n", g_wide_values);
#ifdef SK_EC_COUNT_PASSES
    printf("flat loop passes %lu, codeword passes %lu\n", sk_ec::g_flat_passes, sk_ec::g_flat_codewords);
#endif
    return 0;
}

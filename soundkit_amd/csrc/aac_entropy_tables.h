// aac_entropy_tables.h -- the front-end's tables flattened for aac_entropy_core.h (host or device copies).
// Built by aac_frontend.cpp from the same Lut / Tuple / band-offset data its own parser uses.
#pragma once
#include <stdint.h>

#include <vector>

namespace sk_ec {

constexpr uint32_t kHostPow43Len = (1u << 17) + 64;  // = kPow43Len of aac_entropy_core.h (engine.cpp asserts it)

struct HostTables {
    std::vector<uint32_t> lut;      // all twelve two-level Huffman tables back to back
    uint32_t lut_offset[12];        // [0] scalefactors, [1..11] spectral books
    uint32_t primary_bits[12];
    std::vector<uint64_t> tuples;   // per spectral symbol: bytes 0-3 values, byte 4 sign-bit count, byte 5 escape flag
    uint32_t tuple_offset[12];
    std::vector<float> pow43;       // kPow43Len: every magnitude an escape sequence (+ pulses) can reach
    std::vector<float> sf_wide;     // 65536: scale factors -32768..32767 (the i16 the reference accumulates in)
    std::vector<float> is_wide;     // 65536: intensity positions -32768..32767
    std::vector<float> sf_mult;     // 768
    std::vector<float> is_mult;     // 512: intensity positions -256..255
    std::vector<float> tns_sin;     // 2 x 17
    std::vector<uint16_t> swb;      // every band-offset table back to back
    uint32_t swb_long_offset[13], swb_short_offset[13];
    uint8_t bands_long[13], bands_short[13];
    uint8_t tns_max_long[13], tns_max_short[13];
    std::vector<uint32_t> meta;     // the index block of sk_ec::Tables (aac_entropy_core.h MetaIndex)
};

const HostTables &host_tables();  // aac_frontend.cpp

}  // namespace sk_ec

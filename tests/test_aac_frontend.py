"""AAC-LC access-unit front-end (host code, csrc/aac_frontend.cpp): the reference's own bitstream unit
vectors (soundkit-aac-lc/src/decoder.rs:445-753, config.rs tests), codebook self-checks, robustness
(tests/malformed_decode.rs) and end-to-end checks on real ADTS fixtures.  The GPU-marked tests chain the
front-end into the HIP synthesis; everything else runs on the CPU (the oracle does the synthesis there)."""
import os
import re
from fractions import Fraction

import numpy as np
import pytest

from soundkit_amd import aac_lc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def build_bits(fields):
    """decoder.rs:738-753"""
    bits = []
    for value, width in fields:
        bits += [(value >> b) & 1 for b in range(width - 1, -1, -1)]
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i // 8] |= 1 << (7 - i % 8)
    return bytes(out)


SILENT_SCE = [(0, 3), (0, 4), (100, 8), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]


# ---- AudioSpecificConfig (config.rs:139-319 and its tests) ----------------------------------------------
def test_asc_known_configs():
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x10]))       # decoder.rs:446-456
    assert (fe.sample_rate, fe.channels) == (44100, 2)
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x08]))
    assert (fe.sample_rate, fe.channels) == (44100, 1)
    fe = aac_lc.AacLcFrontEnd(bytes([0x11, 0x90]))       # aac-wasm-bench lib.rs:1884
    assert (fe.sample_rate, fe.channels) == (48000, 2)


@pytest.mark.parametrize("asc,kind", [
    (bytes([0x2B, 0x92, 0x08, 0x00]), "UnsupportedFeature"),            # AOT 5 (SBR) wrapper -> SBR/HE-AAC
    (bytes([0x0A, 0x10]), "UnsupportedAudioObjectType"),                # AOT 1 (Main)
    (bytes([0x12, 0x00]), "UnsupportedFeature"),                        # channel config 0 (PCE)
    (bytes([0x12, 0x18]), "UnsupportedChannelConfig"),                  # channel config 3
    (bytes([0x16, 0x90]), "UnsupportedSamplingFrequencyIndex"),         # frequency index 13
    (bytes([0x12, 0x14]), "UnsupportedFeature"),                        # frame_length_flag -> 960
    (bytes([0x00, 0x00]), "InvalidAudioObjectType"),
    (bytes([0x12]), "UnexpectedEof"),
])
def test_asc_rejections(asc, kind):
    with pytest.raises(aac_lc.AacLcError) as exc:
        aac_lc.AacLcFrontEnd(asc)
    assert exc.value.kind == kind


# ---- access-unit vectors (decoder.rs:481-736) -------------------------------------------------------------
def test_raw_access_unit_entrypoint_reads_element_id():      # decoder.rs:481-492
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x10]))
    with pytest.raises(aac_lc.AacLcError) as exc:
        fe.parse(bytes([0b00100000]))
    assert exc.value.kind == "UnexpectedEof" and "requested 8 bits, 0 bits remain" in str(exc.value)


@pytest.mark.parametrize("extra", [[], [(7, 3)], [(6, 3), (1, 4), (0, 8), (7, 3)]])
def test_silent_sce_with_end_and_fill(extra):                # decoder.rs:494-538
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x08]))
    coeffs, seq, shape = fe.parse(build_bits(SILENT_SCE + extra))
    assert coeffs.shape == (1, 1024) and not coeffs.any() and seq == [0] and shape == [0]


def test_rejects_sbr_fill_and_second_element():              # decoder.rs:540-574
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x08]))
    with pytest.raises(aac_lc.AacLcError) as exc:
        fe.parse(build_bits(SILENT_SCE + [(6, 3), (1, 4), (13, 4), (0, 4)]))
    assert exc.value.kind == "UnsupportedFeature" and "SBR/HE-AAC extension payload" in str(exc.value)
    with pytest.raises(aac_lc.AacLcError) as exc:
        fe.parse(build_bits(SILENT_SCE + [(0, 3), (1, 4)]))
    assert exc.value.kind == "InvalidBitstream" and "multiple channel elements" in str(exc.value)


def test_nonzero_sce_codebook1_tuple():                      # decoder.rs:576-604: tuple [0,0,0,1], sf 100 -> 1.0
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x08]))
    au = build_bits([(0, 3), (0, 4), (100, 8), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (1, 4), (1, 5), (0, 1), (0, 1),
                     (0, 1), (0, 1), (0b10100, 5)])
    coeffs, _, _ = fe.parse(au)
    want = np.zeros(1024, np.float32)
    want[3] = 1.0
    assert np.array_equal(coeffs[0], want)


def test_silent_cpe_and_non_common_window_pair():            # decoder.rs:606-692
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x10]))
    cpe = [(1, 3), (0, 4), (1, 1), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 2)]
    chan = [(100, 8), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]
    coeffs, seq, shape = fe.parse(build_bits(cpe + chan + chan))
    assert coeffs.shape == (2, 1024) and not coeffs.any() and seq == [0, 0]
    fe48 = aac_lc.AacLcFrontEnd(bytes([0x11, 0x90]))
    fields = [(1, 3), (0, 4), (0, 1), (100, 8), (0, 1), (2, 2), (0, 1), (1, 4), (0, 7)]
    fields += [(0, 4), (1, 3)] * 8
    fields += [(0, 1), (0, 1), (0, 1), (100, 8), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]
    coeffs, seq, shape = fe48.parse(build_bits(fields))
    assert not coeffs.any() and seq == [2, 0]


def test_cpe_spectral_data_before_next_channel_header():     # decoder.rs:694-736
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x10]))
    fields = [(1, 3), (0, 4), (1, 1), (0, 1), (0, 2), (0, 1), (1, 6), (0, 1), (0, 2),
              (100, 8), (1, 4), (1, 5), (0, 1), (0, 1), (0, 1), (0, 1), (0b10100, 5),
              (100, 8), (0, 4), (1, 5), (0, 1), (0, 1), (0, 1)]
    coeffs, _, _ = fe.parse(build_bits(fields))
    assert coeffs[0, 3] == 1.0 and np.count_nonzero(coeffs[0]) == 1 and not coeffs[1].any()


def test_gain_control_prediction_and_reserved_rejections():
    fe = aac_lc.AacLcFrontEnd(bytes([0x12, 0x08]))
    bad_gain = SILENT_SCE[:-1] + [(1, 1)]                        # channel.rs:66-69
    with pytest.raises(aac_lc.AacLcError) as exc:
        fe.parse(build_bits(bad_gain))
    assert exc.value.kind == "UnsupportedFeature" and "gain control" in str(exc.value)
    pred = SILENT_SCE[:7] + [(1, 1)]                             # ics.rs:91-94
    with pytest.raises(aac_lc.AacLcError) as exc:
        fe.parse(build_bits(pred + SILENT_SCE[8:]))
    assert "AAC prediction" in str(exc.value)
    cb12 = SILENT_SCE[:8] + [(12, 4)] + SILENT_SCE[9:]           # section.rs:27-29
    with pytest.raises(aac_lc.AacLcError) as exc:
        fe.parse(build_bits(cb12))
    assert "reserved AAC section codebook" in str(exc.value)
    with pytest.raises(aac_lc.AacLcError) as exc:                # decoder.rs:134-145
        fe.parse(build_bits([(2, 3), (0, 4)]))
    assert "channel coupling element" in str(exc.value)
    with pytest.raises(aac_lc.AacLcError) as exc:                # decoder.rs:151-155
        fe.parse(build_bits([(7, 3)]))
    assert "does not contain an AAC-LC channel element" in str(exc.value)
    with pytest.raises(aac_lc.AacLcError) as exc:                # decoder.rs:157-161
        fe.parse(build_bits(SILENT_SCE + [(7, 3)]) + b"\x01")
    assert "non-zero trailing bits" in str(exc.value)


# ---- codebooks: complete prefix codes (as spectral.rs:2575+ checks) ---------------------------------------
def test_huffman_codebooks_are_complete_prefix_codes():
    text = open(os.path.join(os.path.dirname(HERE), "soundkit_amd", "csrc", "aac_tables.h")).read()

    def arr(name):
        body = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, text, re.S).group(1)
        return [int(x) for x in re.findall(r"\d+", body)]
    sizes = {"Sf": 121, "Cb1": 81, "Cb2": 81, "Cb3": 81, "Cb4": 81, "Cb5": 81, "Cb6": 81, "Cb7": 64, "Cb8": 64, "Cb9": 169,
             "Cb10": 169, "Cb11": 289}
    for book, n in sizes.items():
        lens, codes = arr("k%sLen" % book), arr("k%sCode" % book)
        assert len(lens) == len(codes) == n
        assert sum(Fraction(1, 2 ** l) for l in lens) == 1, book          # Kraft equality
        words = sorted(format(c, "0%db" % l) for l, c in zip(lens, codes))
        assert all(not b.startswith(a) for a, b in zip(words, words[1:])), book


# ---- ADTS framing (soundkit-decoder lib.rs:1007-1027) ------------------------------------------------------
def test_parse_adts_access_unit():
    data = open(os.path.join(GOLD, "aac", "stereo-music-44100-192k.aac"), "rb").read()
    asc, au, flen = aac_lc.parse_adts_access_unit(data)
    assert asc == bytes([0x12, 0x10]) and len(au) == flen - 7
    frames = aac_lc.split_adts(data)
    assert len(frames) == 131 and all(a == asc for a, _ in frames)      # 131 ADTS frames
    for bad in (b"", b"\xff\xf1\x50", b"\x00" * 16, b"\xff\x00" + b"\x00" * 14):
        with pytest.raises(ValueError):
            aac_lc.parse_adts_access_unit(bad)


def with_crc(data):
    """The same ADTS stream with protection_absent = 0: two CRC bytes after each 7-byte header (not verified by the
    reference's parse_adts_access_unit either, lib.rs:1007-1027), frame length + 2."""
    out = bytearray()
    pos = 0
    while pos + 7 <= len(data):
        flen = ((data[pos + 3] & 3) << 11) | (data[pos + 4] << 3) | (data[pos + 5] >> 5)
        h = bytearray(data[pos:pos + 7])
        h[1] &= 0xFE
        n = flen + 2
        h[3] = (h[3] & 0xFC) | ((n >> 11) & 3)
        h[4] = (n >> 3) & 0xFF
        h[5] = (h[5] & 0x1F) | ((n & 7) << 5)
        out += h + b"\xAB\xCD" + data[pos + 7:pos + flen]
        pos += flen
    return bytes(out)


def test_adts_with_crc_header():
    data = open(os.path.join(GOLD, "aac", "aac-stereo-48k.adts"), "rb").read()
    plain, crc = aac_lc.split_adts(data), aac_lc.split_adts(with_crc(data))
    assert len(plain) == len(crc) == 48 and all(a == b for a, b in zip(plain, crc))


# ---- robustness: soundkit-aac-lc/tests/malformed_decode.rs:5-39 ----------------------------------------------
@pytest.mark.parametrize("asc", [bytes([0x12, 0x10]), bytes([0x11, 0x88])])
def test_malformed_access_units_never_crash(asc):
    fe = aac_lc.AacLcFrontEnd(asc)
    state = 0x9E3779B97F4A7C15
    ok = 0
    for i in range(2048):
        n = 1 + (i * 37) % 700
        buf = bytearray(n)
        for k in range(n):
            state ^= (state << 13) & 0xFFFFFFFFFFFFFFFF
            state ^= state >> 7
            state ^= (state << 17) & 0xFFFFFFFFFFFFFFFF
            buf[k] = state & 0xFF
        try:
            coeffs, _, _ = fe.parse(bytes(buf))
            assert np.isfinite(coeffs).all()
            ok += 1
        except aac_lc.AacLcError:
            pass
    assert ok < 2048  # random bytes are overwhelmingly rejected, never fatal


def test_mutated_access_units_under_sanitizers(tmp_path):
    """Bit flips, byte overwrites, truncations and insertions applied to real access units (so the mutants reach
    the deep paths: sections, escapes, PNS, intensity, TNS), front-end built with ASan + UBSan."""
    import subprocess
    exe = str(tmp_path / "fuzz_frontend")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-Wno-subobject-linkage", "-o", exe, os.path.join(HERE, "fuzz_frontend.cpp")],
                          cwd=HERE)
    files = [os.path.join(GOLD, "aac", n) for n in ("aac-stereo-48k.adts", "stereo-music-44100-192k.aac", "mono16k_A_Tusk.aac")]
    out = subprocess.run([exe, "12000"] + files, capture_output=True, text=True, cwd=HERE)
    assert out.returncode == 0, out.stderr[-2000:]
    ok, err = [int(x) for x in re.findall(r"ok (\d+) err (\d+)", out.stdout)[0]]
    assert ok > 0 and err > 0 and ok + err == 36000


def test_no_heap_allocation_after_warm_up(tmp_path):
    """soundkit-aac-lc/tests/no_alloc_decode.rs: decoding must not allocate once the decoder is warm."""
    import subprocess
    exe = str(tmp_path / "noalloc_frontend")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-subobject-linkage", "-o", exe,
                           os.path.join(HERE, "noalloc_frontend.cpp")], cwd=HERE)
    out = subprocess.run([exe] + [os.path.join(GOLD, "aac", n) for n in FIXTURES], capture_output=True, text=True)
    assert out.returncode == 0 and "273 frames, 0 allocations after warm-up" in out.stdout, out.stdout


# ---- real fixtures -----------------------------------------------------------------------------------------
# aac-stereo-48k.adts = the elementary stream of testdata/mpeg-ts/aac-stereo-48k.ts (tools/ts_to_adts.py)
FIXTURES = ["stereo-music-44100-192k.aac", "A_Tusk_is_used_to_make_costly_gifts_encoded.aac", "mono16k_A_Tusk.aac",
            "aac-stereo-48k.adts"]


def parse_file(name):
    frames = aac_lc.split_adts(open(os.path.join(GOLD, "aac", name), "rb").read())
    fe = aac_lc.AacLcFrontEnd(frames[0][0])
    out = [fe.parse(au) for _, au in frames]
    return fe, out


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_parses_to_the_last_bit(name):
    """Every access unit must be consumed exactly (the trailing-zero check of decoder.rs:157-161 passes):
    any error in the codebooks or the syntax would desynchronise the reader within a frame or two."""
    fe, out = parse_file(name)
    assert len(out) == {"stereo-music-44100-192k.aac": 131, "A_Tusk_is_used_to_make_costly_gifts_encoded.aac": 46,
                        "mono16k_A_Tusk.aac": 48, "aac-stereo-48k.adts": 48}[name]
    assert all(np.isfinite(c).all() for c, _, _ in out)


def test_fixture_tool_coverage():
    """aac-wasm-bench/src/lib.rs:1955-1986 asserts short windows, TNS, PNS, IS and MS on its fixture."""
    total = {}
    for name in FIXTURES:
        fe, _ = parse_file(name)
        for k, v in fe.tool_usage().items():
            total[k] = total.get(k, 0) + v
    for tool in ("short", "transition", "tns", "pns_bands", "is_bands", "ms_bands"):
        assert total[tool] > 0, tool


def decode_with(synth, name):
    fe, out = parse_file(name)
    return fe, np.concatenate([synth(fe, c, s, sh) for c, s, sh in out], axis=1)


def snr_against_source(decoded_left, source_left):
    best = (-99.0, 0)
    for lag in (1024, 2048, 2049, 2112, 3072):
        n = min(decoded_left.size - lag, source_left.size)
        a, b = decoded_left[lag:lag + n].astype(np.float64), source_left[:n].astype(np.float64)
        g = (a * b).sum() / (b * b).sum()
        err = ((a - g * b) ** 2).sum()
        best = max(best, (10 * np.log10((a * a).sum() / err), lag))
    return best


def source_wav():
    from test_pcm_gpu import read_wav
    _, pcm = read_wav(os.path.join(GOLD, "wav_stereo_A_Tusk.wav"))
    return np.frombuffer(pcm, "<i2").reshape(-1, 2).T.astype(np.float32) / 32768.0


def test_decoded_fixture_reproduces_its_source_waveform(oracle):
    """golden/aac/A_Tusk..._encoded.aac was encoded from testdata/wav_stereo/A_Tusk....wav: decoding it
    (front-end + oracle synthesis) must give that waveform back (lossy: > 25 dB SNR at the encoder delay).
    This frame set uses TNS, mid/side, eight-short and transition windows."""
    chans = {}

    def synth(fe, c, s, sh):
        st = chans.setdefault(id(fe), [oracle.Channel() for _ in range(fe.channels)])
        pcm, _ = oracle.synthesize_stream(c[None], [s + [0] * (2 - len(s))], [sh + [0] * (2 - len(sh))], st)
        return pcm[0]
    _, dec = decode_with(synth, "A_Tusk_is_used_to_make_costly_gifts_encoded.aac")
    snr, lag = snr_against_source(dec[0], source_wav()[0])
    assert lag == 2048 and snr > 25.0, (snr, lag)
    assert np.abs(dec[0] - dec[1]).max() < 1e-6  # dual-mono source: L == R through the stereo tools


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_synthesis_of_real_spectra_matches_oracle(engine, oracle, name):
    """SURVEY config 2: the spectra the front-end extracts from real AAC run through the HIP kernel and
    through the oracle; 1e-6 RMS."""
    fe, out = parse_file(name)
    ch = fe.channels
    coeffs = np.stack([c for c, _, _ in out])
    seqs = np.array([s + [0] * (2 - ch) for _, s, _ in out], np.uint8)
    shapes = np.array([sh + [0] * (2 - ch) for _, _, sh in out], np.uint8)
    sid = engine.open_stream(fe.sample_rate, ch)
    pcm, status = aac_lc.synthesize_batch(engine, [sid] * len(out), ch, coeffs, seqs, shapes)
    engine.close_stream(sid)
    want, _ = oracle.synthesize_stream(coeffs, seqs, shapes)
    assert not status.any()
    err = np.sqrt(np.mean((pcm.astype(np.float64) - want) ** 2)) / np.sqrt(np.mean(want.astype(np.float64) ** 2))
    assert err < 1e-6, err
    assert np.abs(pcm - want).max() < 2e-6 * np.abs(want).max()


@pytest.mark.gpu
def test_access_unit_decoder_mirror_end_to_end(engine, oracle):
    """AacLcDecoder-shaped object: ADTS frame in, PlanarF32 / interleaved i16 out, entropy on the host and
    synthesis on the GPU; the result reproduces the source waveform."""
    frames = aac_lc.split_adts(open(os.path.join(GOLD, "aac", "A_Tusk_is_used_to_make_costly_gifts_encoded.aac"), "rb").read())
    dec = aac_lc.AacLcDecoder.from_audio_specific_config(frames[0][0], engine)
    info = dec.frame_info()
    assert (info.sample_rate, info.channels, info.frames) == (16000, 2, 1024)
    planar = np.concatenate([dec.decode_access_unit(au).channels() for _, au in frames], axis=1)
    snr, lag = snr_against_source(planar[0], source_wav()[0])
    assert lag == 2048 and snr > 25.0
    dec2 = aac_lc.AacLcDecoder(frames[0][0], engine)
    s16 = np.concatenate([dec2.decode_access_unit_s16(au).reshape(1024, 2) for _, au in frames])
    assert np.array_equal(s16.ravel(), oracle.planar_f32_to_s16_interleaved(planar))
    with pytest.raises(aac_lc.AacLcError):
        dec.decode_access_unit(b"\xff\xff\xff")
    dec.close(), dec2.close()

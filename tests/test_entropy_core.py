"""csrc/aac_entropy_core.h -- the front-end source that also compiles for gfx950 -- built for the CPU with
AddressSanitizer + UBSan and proved equal to the host front-end (csrc/aac_frontend.cpp): same status code,
bit-identical spectra, same window fields and PNS generator state on all fixture access units and on mutated ones.
(A 1.6-million-unit run of the same harness was clean when the core was written.)  The GPU build of the same source is
then checked against the host path in tests/test_entropy_gpu.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["aac-stereo-48k.adts", "stereo-music-44100-192k.aac", "mono16k_A_Tusk.aac", "A_Tusk_is_used_to_make_costly_gifts_encoded.aac"]


# the spectral decode exists in two forms (aac_entropy_core.h): nested loops (host builds) and one flat loop (the device
# build; -DSK_EC_FLAT selects it on the host) -- both are checked
FORMS = [pytest.param([], id="nested"), pytest.param(["-DSK_EC_FLAT"], id="flat")]


@pytest.mark.parametrize("form", FORMS)
def test_entropy_core_equals_host_front_end_under_sanitizers(tmp_path, form):
    exe = str(tmp_path / "entropy_core_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-Wno-subobject-linkage"] + form + ["-o", exe, os.path.join(HERE, "entropy_core_check.cpp")],
                          cwd=HERE)
    out = subprocess.run([exe, "6000"] + [os.path.join(HERE, "golden", "aac", f) for f in FILES], capture_output=True, text=True)
    assert out.returncode == 0 and "identical" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    assert "checked 24273 access units" in out.stdout


@pytest.mark.parametrize("form", FORMS)
def test_entropy_core_equals_host_front_end_on_generated_units(tmp_path, form):
    """the same harness on units from tests/au_builder.py (pulse data, escapes up to the 16-bit limit, every codebook and
    window grouping, values beyond the reference's tables) and on 300 mutants of each generated stream"""
    import numpy as np

    from au_builder import adts_frame, extreme_scalefactor_units, random_access_unit
    exe = str(tmp_path / "entropy_core_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-Wno-subobject-linkage"] + form + ["-o", exe, os.path.join(HERE, "entropy_core_check.cpp")],
                          cwd=HERE)
    files, units = [], 0
    for k, (sf_index, channels) in enumerate([(3, 2), (4, 1), (8, 2), (11, 1), (0, 2)]):
        rng = np.random.default_rng(900 + k)
        aus = [random_access_unit(rng, sf_index, channels) for _ in range(60)]
        aus += [au for sf, ch, au in extreme_scalefactor_units() if (sf, ch) == (sf_index, channels)]
        path = str(tmp_path / ("gen%d.adts" % k))
        open(path, "wb").write(b"".join(adts_frame(au, sf_index, channels) for au in aus))
        files.append(path)
        units += len(aus) + 300
    out = subprocess.run([exe, "300"] + files, capture_output=True, text=True)
    assert out.returncode == 0 and "identical" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    assert "checked %d access units" % units in out.stdout


def test_quantised_hand_over_carries_values_beyond_i16(tmp_path):
    """escape sequences of up to 16 extra bits (magnitudes to 131071, spectral.rs:214-228): the quantised hand-over's list
    of wide values must reproduce the f32 path bit for bit (host build of the core, sanitizers on)"""
    import re

    import numpy as np

    import au_builder
    from au_builder import adts_frame, random_access_unit
    exe = str(tmp_path / "entropy_core_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-Wno-subobject-linkage", "-o", exe, os.path.join(HERE, "entropy_core_check.cpp")], cwd=HERE)
    saved = au_builder.ESCAPE_SIZES
    au_builder.ESCAPE_SIZES = [0, 4, 9, 11, 12]
    try:
        files = []
        for k, (sf_index, channels) in enumerate([(3, 2), (4, 1), (8, 2)]):
            rng = np.random.default_rng(1900 + k)
            aus = [random_access_unit(rng, sf_index, channels) for _ in range(60)]
            path = str(tmp_path / ("wide%d.adts" % k))
            open(path, "wb").write(b"".join(adts_frame(au, sf_index, channels) for au in aus))
            files.append(path)
    finally:
        au_builder.ESCAPE_SIZES = saved
    out = subprocess.run([exe, "100"] + files, capture_output=True, text=True)
    assert out.returncode == 0 and "identical" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    assert int(re.search(r"wide values (\d+)", out.stdout).group(1)) > 50, out.stdout[-200:]

"""The arithmetic budget of fir_bf16.hip on the host (tests/fir_split_model.py): the products its kProducts table leaves
out must not show against an f64 evaluation, and the table in the kernel source must be the one the model checks."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_left_out_products_are_below_f32_rounding():
    import fir_split_model as model
    kept = model.SETS["41 (kProducts)"]
    src = open(os.path.join(HERE, "..", "soundkit_amd", "csrc", "fir_bf16.hip")).read()
    table = re.search(r"kProducts\[kWindows\] = \{([0-9, ]+)\}", src).group(1)
    assert [int(v) for v in table.split(",")] == kept
    err = model.errors({k: model.SETS[k] for k in ("60 MFMAs per tile (all six products everywhere)", "41 (kProducts)")}, n=12000)
    full, cut = err["60 MFMAs per tile (all six products everywhere)"], err["41 (kProducts)"]
    assert full[0] < 1.6e-7 and cut[0] < 1.6e-7 and cut[0] < full[0] * 1.05  # relative RMS against f64
    assert cut[1] < 1e-6                                                     # max abs, full-scale input


def test_f16_planes_of_s16_rows_stay_inside_the_bound():
    """the f16 form (two planes per operand, products x1h1 | x1h2, x2h1, 24 per tile): <= 2e-7 relative RMS against f64 at
    full scale and at small amplitudes alike (the remainder plane scales with the sample, not with full scale)"""
    import fir_split_model as model
    # the table is shared by fir_bf16.hip and the fused decode-tail kernel: it lives in sk_device.h
    src = open(os.path.join(HERE, "..", "soundkit_amd", "csrc", "sk_device.h")).read()
    table = re.search(r"kFirProductsF16\[10\] = \{([0-9, ]+)\}", src).group(1)
    assert [int(v) for v in table.split(",")] == model.F16_SETS["f16, 24 (kProductsF16)"]
    r = model.errors_f16(n=12000)
    for (label, amp), (rms, _) in r.items():
        assert rms < 2.0e-7, (label, amp, rms)
    kept = {amp: rms for (label, amp), (rms, _) in r.items() if "kProductsF16" in label}
    full = {amp: rms for (label, amp), (rms, _) in r.items() if "all four" in label}
    for amp in kept:
        assert kept[amp] < 1.1 * full[amp] + 1e-9   # what kProductsF16 leaves out does not show

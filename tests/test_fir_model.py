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

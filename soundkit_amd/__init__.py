"""soundkit_amd -- MI355X (gfx950) batched decode-DSP engine behind soundkit's decoder surface.

The product is the HIP library libsoundkit_amd.so (C ABI: include/soundkit_amd.h).  This
package is its Python-side host mirror: the same names, argument meaning and error
behaviour as the reference's Rust surfaces for this path, so parity tests read like the
reference's own tests.  Importing it without the built library raises ImportError; there
is no CPU fallback.
"""
from ._lib import LIB_PATH, SoundkitError, declared_symbols, lib  # noqa: F401
from .engine import (EIGHT_SHORT, KBD, LONG_START, LONG_STOP, ONLY_LONG, SINE, Engine, Plan,  # noqa: F401
                     default_engine, descs_from_arrays, make_descs)

__version__ = "0.1.0"

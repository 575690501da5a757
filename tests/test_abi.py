"""CPU-side checks of the C-ABI: the libraries load and export every symbol the headers declare,
the ctypes signature tables cover them, and argument validation fails with a message (no GPU use)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_vitk_exports_every_declared_symbol():
    from visiontransformer_amd import vitk
    names = declared("vitk.h", "vitk_")
    assert len(names) >= 25
    d = vitk.dll()
    for n in names:
        assert hasattr(d, n), f"libvitk.so does not export {n}"
        assert n in vitk.SIGNATURES, f"no ctypes signature for {n}"
    assert sorted(vitk.SIGNATURES) == names
    assert d.vitk_version() >= 100


def test_vitk_argument_validation_reports_errors():
    from visiontransformer_amd import vitk
    d = vitk.dll()
    a = vitk.GemmArgs()
    a.M, a.N, a.K = 4, 4, 4  # null pointers
    assert d.vitk_gemm_f32(C.byref(a), None) != 0
    assert b"null operand" in d.vitk_last_error()
    with pytest.raises(vitk.VitkError):
        vitk.call("vitk_layernorm_fwd", None, 4, 8, 4, None, None, 1e-5, None, None, None, None)  # stride < D
    with pytest.raises(vitk.VitkError):
        vitk.call("vitk_dropout_fwd", None, 4, 1.5, 0, 0, None, None, None)
    assert C.sizeof(vitk.GemmArgs) == 200

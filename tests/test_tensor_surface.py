"""The public Tensor API of the reference (include/core/Tensor.hpp:23-98 -- views, sum, square,
addBroadcast, operator+, contiguous, expand, concatenate, matrixMultiply, batchMatrixMultiply,
element access) on the device-resident Tensor of the host layer.

tests/tensor_surface.inc holds 30 scenarios written once against that API and compiled twice,
unchanged: against the reference (oracle/_ref, which produced tests/golden/tensor_surface.npz) and
against this repo's host layer (tests/native/libtensor_surface.so).  The CPU part pins the golden
values to a numpy restatement of each scenario; the GPU part runs the host-layer build."""
import ctypes as C
import os

import numpy as np
import pytest

import vitlibs as V


def numpy_scenarios():
    a, b = V.tensor_surface_inputs()
    a, b = a.astype(np.float64), b.astype(np.float64)

    def mk(shape, src=a, off=0):
        n = int(np.prod(shape))
        return src[off:off + n].reshape(shape)

    s = {}
    s[0] = mk((7, 5)).sum(0, keepdims=True)
    s[1] = mk((7, 5)).sum(1, keepdims=True)
    s[2] = mk((3, 4, 5)).sum(0, keepdims=True)
    s[3] = mk((3, 4, 5)).sum(1, keepdims=True)
    s[4] = mk((3, 4, 5)).sum(2, keepdims=True)
    s[5] = mk((7, 5)).T.sum(0, keepdims=True)
    s[6] = mk((6, 9, 4))[:, 2:7].sum(0, keepdims=True)
    s[7] = mk((6, 11)) ** 2
    s[8] = mk((2, 5, 7)) ** 2
    s[9] = mk((9, 13)) + mk((1, 13), b)
    s[10] = mk((4, 5, 6)) + mk((1, 5, 6), b)
    s[11] = mk((8, 12)) + mk((8, 12), b)
    s[12] = mk((3, 5, 7)) + mk((3, 5, 7), b)
    s[13] = mk((8, 12)).T + mk((12, 8), b)
    s[14] = mk((3, 5, 7)).transpose(0, 2, 1) + mk((3, 7, 5), b)
    s[15] = mk((3, 5, 7)).transpose(0, 2, 1)
    s[16] = mk((2, 6, 12)).reshape(2, 6, 3, 4).transpose(0, 2, 1, 3)
    s[17] = mk((2, 3, 6, 4)).transpose(0, 2, 1, 3).reshape(2, 6, 12)
    s[18] = mk((5, 6, 4))[1:4]
    s[19] = np.broadcast_to(mk((1, 1, 8)), (5, 1, 8))
    s[20] = np.concatenate([np.broadcast_to(mk((1, 1, 8)), (3, 1, 8)), mk((3, 4, 8), b)], 1) + mk((1, 5, 8), b, 200)
    s[21] = np.concatenate([mk((2, 3, 4)), mk((5, 3, 4), b)], 0)
    s[22] = np.concatenate([mk((2, 3, 4)), mk((2, 3, 6), b), mk((2, 3, 1), a, 100)], 2)
    s[23] = np.broadcast_to(mk((4, 6)), (3, 4, 6))
    s[24] = mk((9, 14)) @ mk((14, 6), b)
    s[25] = mk((14, 9)).T @ mk((6, 14), b).T
    s[26] = mk((4, 5, 8)) @ mk((4, 8, 6), b)
    s[27] = mk((4, 5, 8)) @ mk((4, 5, 8), b).transpose(0, 2, 1)
    x = np.full((4, 6), 0.5)
    x[np.arange(4), np.arange(4)] = a[:4]
    y = x.copy()
    y[0, 5] = -3.0
    s[28] = x @ mk((6, 3), b) + y @ mk((6, 3), b)
    x = mk((3, 4, 5)).copy()
    x[0, 1, 0] = 42.0
    s[29] = x.sum(1, keepdims=True)
    return s


def test_golden_values_of_the_reference_match_a_numpy_restatement():
    g = V.golden("tensor_surface.npz")
    want = numpy_scenarios()
    assert len(want) == V.TENSOR_SCENARIOS
    for sc in range(V.TENSOR_SCENARIOS):
        got = g[f"s{sc}"]
        assert got.shape == want[sc].shape, (sc, got.shape, want[sc].shape)
        assert V.max_relerr(got, want[sc]) <= 2e-6, sc


@pytest.mark.skipif(V.ref() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_golden_values_are_what_the_reference_build_gives():
    g = V.golden("tensor_surface.npz")
    for sc in range(V.TENSOR_SCENARIOS):
        np.testing.assert_array_equal(V.run_tensor_scenario(V.ref().dll.vref_tensor_scenario, sc), g[f"s{sc}"])


@pytest.mark.gpu
def test_host_layer_tensor_api_matches_the_reference():
    from visiontransformer_amd import host
    host.init(0)
    path = os.path.join(V.ROOT, "tests", "native", "libtensor_surface.so")
    assert os.path.exists(path), "build it with `make -C tests/native` (or __graft_entry__.build())"
    fn = C.CDLL(path).vith_test_tensor_scenario
    fn.restype, fn.argtypes = C.c_int, [C.c_int] + [C.c_void_p] * 5
    g = V.golden("tensor_surface.npz")
    for sc in range(V.TENSOR_SCENARIOS):
        got = V.run_tensor_scenario(fn, sc)
        want = g[f"s{sc}"]
        assert got.shape == want.shape, (sc, got.shape, want.shape)
        if sc in (15, 16, 17, 18, 19, 21, 22, 23):  # pure data movement
            np.testing.assert_array_equal(got, want, err_msg=f"scenario {sc}")
        else:
            V.assert_close(got, want, 2e-6, f"scenario {sc}")

"""ctypes binding of include/vitk.h (lib/libvitk.so).  Arguments may be torch CUDA tensors
(float32/int32, contiguous unless a pitch is passed separately), raw integer addresses or None."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvitk.so")

ACT_NONE, ACT_GELU_FWD, ACT_GELU_BWD = 0, 1, 2
DK_REFERENCE, DK_EXACT = 0, 1
PATCH_REFERENCE, PATCH_IM2COL = 0, 1


class VitkError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("trans_a", C.c_int32), ("trans_b", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("alpha", C.c_float), ("act", C.c_int32),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("batch_outer", C.c_int32), ("batch_inner", C.c_int32),
        ("stride_a_outer", C.c_int64), ("stride_a_inner", C.c_int64),
        ("stride_b_outer", C.c_int64), ("stride_b_inner", C.c_int64),
        ("stride_c_outer", C.c_int64), ("stride_c_inner", C.c_int64),
        ("split_k", C.c_int32), ("reserved", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("colsum_b", C.c_void_p),
    ]


_p, _i32, _i64, _u64, _f, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every function of include/vitk.h
SIGNATURES = {
    "vitk_version": (C.c_int, []),
    "vitk_last_error": (C.c_char_p, []),
    "vitk_init": (C.c_int, [C.c_int]),
    "vitk_gemm_f32": (C.c_int, [C.POINTER(GemmArgs), _p]),
    "vitk_gemm_pair_f32": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _p]),
    "vitk_gemm_workspace_bytes": (_sz, [C.POINTER(GemmArgs)]),
    "vitk_gemm_profile_enable": (C.c_int, [C.c_int]),
    "vitk_gemm_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_int64), C.c_int]),
    "vitk_gemm_variant_name": (C.c_char_p, [C.c_int]),
    "vitk_gemm_small_enable": (C.c_int, [C.c_int]),
    "vitk_set_planning_cu_count": (C.c_int, [C.c_int]),
    "vitk_sdpa_tile16_enable": (C.c_int, [C.c_int]),
    "vitk_sdpa_bwd_tile16_enable": (C.c_int, [C.c_int]),
    "vitk_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "vitk_colsum_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "vitk_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "vitk_layernorm_fwd": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _f, _p, _p, _p, _p]),
    "vitk_layernorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "vitk_layernorm_bwd_static_max_d": (_sz, []),
    "vitk_layernorm_small_enable": (C.c_int, [C.c_int]),
    "vitk_layernorm_bwd_static": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p]),
    "vitk_gelu_fwd": (C.c_int, [_p, _i64, _p, _p]),
    "vitk_gelu_bwd": (C.c_int, [_p, _p, _i64, _p, _p]),
    "vitk_dropout_fwd": (C.c_int, [_p, _i64, _f, _u64, _u64, _p, _p, _p]),
    "vitk_dropout_bwd": (C.c_int, [_p, _p, _i64, _p, _p]),
    "vitk_dropout_residual_fwd": (C.c_int, [_p, _p, _i64, _f, C.c_uint64, C.c_uint64, _p, _p, _p]),
    "vitk_dropout_stream_bwd": (C.c_int, [_p, _i64, _f, C.c_uint64, C.c_uint64, _p, _p]),
    "vitk_dropout_rows": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _f, C.c_uint64, C.c_uint64, _p, _p]),
    "vitk_add_f32": (C.c_int, [_p, _p, _i64, _p, _p]),
    "vitk_add3_f32": (C.c_int, [_p, _p, _p, _i64, _p, _p]),
    "vitk_scale_f32": (C.c_int, [_p, _f, _i64, _p, _p]),
    "vitk_fill_f32": (C.c_int, [_p, _f, _i64, _p]),
    "vitk_add_rowvec_f32": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "vitk_copy2d_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "vitk_copy4d_f32": (C.c_int, [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _p, _p]),
    "vitk_softmax_rows": (C.c_int, [_p, _i64, _i64, _i64, _p, _p]),
    "vitk_softmax_rows_bwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _f, _p, _p]),
    "vitk_dk_reference_scatter": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "vitk_gather_rows_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _p]),
    "vitk_augment_f32": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    "vitk_sdpa_head_kernels_enable": (C.c_int, [C.c_int]),
    "vitk_sdpa_supported": (C.c_int, [_i32, _i32]),
    "vitk_sdpa_fwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _f, _p, _i64, _p, _p]),
    "vitk_sdpa_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "vitk_sdpa_bwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _p, _i32, _i32, _i32, _i32, _f, _i32, _p, _p, _p, _i64,
                                _p, _sz, _p]),
    "vitk_sdpa_row_supported": (C.c_int, [_i32, _i32]),
    "vitk_sdpa_row_fwd": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _f, _p, _i64, _p, _p]),
    "vitk_sdpa_row_bwd": (C.c_int, [_p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i32, _i32, _i32, _i32, _f, _i32, _p, _i64,
                                    _p, _p, _i64, _p]),
    "vitk_add2d_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "vitk_patch_gather": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "vitk_patch_scatter_bwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "vitk_embed_assemble": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "vitk_embed_split_bwd": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "vitk_cls_gather": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "vitk_cls_scatter": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "vitk_softmax_ce": (C.c_int, [_p, _p, _i32, _i32, _p, _f, _p, _p, _p, _p, _sz, _p]),
    "vitk_argmax_rows": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "vitk_adam_l2": (C.c_int, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _f, _f, _f, _p]),
}

_dll = None


def dll():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise VitkError(f"{LIB_PATH} is missing: build it with `make -C visiontransformer_amd` "
                            "(or __graft_entry__.build()); there is no CPU fallback")
        d = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(d, name)
            fn.restype, fn.argtypes = res, args
        _dll = d
    return _dll


def ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def check(rc):
    if rc != 0:
        raise VitkError(dll().vitk_last_error().decode())


def call(name, *args):
    """Call a status-returning entry point, converting tensors to addresses and raising on error."""
    fn = getattr(dll(), name)
    conv = [ptr(a) if (at is _p) else a for a, at in zip(args, fn.argtypes)]
    assert len(conv) == len(fn.argtypes), (name, len(conv), len(fn.argtypes))
    check(fn(*conv))


def init(device=0):
    check(dll().vitk_init(device))


def _gemm_args(A, B, Cout, M, N, K, *, trans_a=False, trans_b=False, lda=None, ldb=None, ldc=None, alpha=1.0,
               act=ACT_NONE, bias=None, residual=None, ldr=None, aux=None, ldaux=None, batch_outer=1, batch_inner=1,
               stride_a=(0, 0), stride_b=(0, 0), stride_c=(0, 0), split_k=1, colsum_b=None, tile=None):
    a = GemmArgs()
    a.reserved = {None: 0, "64x64": 1, "128x64": 2, "64x128": 3}[tile]
    a.trans_a, a.trans_b, a.M, a.N, a.K = int(trans_a), int(trans_b), M, N, K
    a.A, a.B, a.C = ptr(A), ptr(B), ptr(Cout)
    a.lda = lda if lda is not None else (M if trans_a else K)
    a.ldb = ldb if ldb is not None else (K if trans_b else N)
    a.ldc = ldc if ldc is not None else N
    a.alpha, a.act = alpha, act
    a.bias, a.residual, a.aux = ptr(bias), ptr(residual), ptr(aux)
    a.ldr = ldr if ldr is not None else N
    a.ldaux = ldaux if ldaux is not None else N
    a.batch_outer, a.batch_inner = batch_outer, batch_inner
    a.stride_a_outer, a.stride_a_inner = stride_a
    a.stride_b_outer, a.stride_b_inner = stride_b
    a.stride_c_outer, a.stride_c_inner = stride_c
    a.split_k = split_k
    a.colsum_b = ptr(colsum_b)
    return a


def _give_workspace(a, workspace):
    """Attach `workspace` (or, for tests, a fresh torch buffer) when these arguments need one."""
    need = dll().vitk_gemm_workspace_bytes(C.byref(a))
    keep = None
    if need:
        if workspace is None:
            import torch
            keep = workspace = torch.empty(need // 4, dtype=torch.float32, device="cuda")
        a.workspace, a.workspace_bytes = ptr(workspace), workspace.numel() * 4
    return keep


def gemm(A, B, Cout, M, N, K, *, workspace=None, st=None, **kw):
    """C = epilogue(alpha * op(A) op(B)); see include/vitk.h.  With split_k != 1 and no workspace given,
    one is allocated with torch (tests only; the host layer passes its own).  tile: None = the
    library's choice; "64x64", "128x64", "64x128" force that output tile (tests, measurements)."""
    a = _gemm_args(A, B, Cout, M, N, K, **kw)
    keep = _give_workspace(a, workspace)
    check(dll().vitk_gemm_f32(C.byref(a), st if st is not None else stream()))
    return keep


def gemm_pair(dx, dw, st=None):
    """vitk_gemm_pair_f32: dx and dw are dicts of gemm() arguments (A, B, Cout, M, N, K, ...) of the two
    products of a Dense backward."""
    ax, aw = _gemm_args(**dx), _gemm_args(**dw)
    keep = (_give_workspace(ax, None), _give_workspace(aw, None))
    check(dll().vitk_gemm_pair_f32(C.byref(ax), C.byref(aw), st if st is not None else stream()))
    return keep


def planning_cu_count(cus):
    """Plans (kernel routes, split counts, tile shapes) as on a device with `cus` CUs; 0 = this device's."""
    check(dll().vitk_set_planning_cu_count(int(cus)))


def gemm_small(enable):
    """Route small problems to the one-wave-per-tile kernel (the default) or keep them on the tiled one."""
    check(dll().vitk_gemm_small_enable(int(enable)))


def sdpa_tile16(enable):
    """Attention (forward and backward) of head dimensions 33..64 on 16-row tiles (the default) or on the 32-row kernels."""
    check(dll().vitk_sdpa_tile16_enable(int(bool(enable))))
    check(dll().vitk_sdpa_bwd_tile16_enable(int(enable)))  # (n >= 2: n heads per workgroup, for tests)


def gemm_profile(enable):
    check(dll().vitk_gemm_profile_enable(int(enable)))


def gemm_profile_read(reset=True):
    """Per kernel variant: dict(name, ms, flops, launches) for variants that were launched."""
    out = []
    for v in range(53):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        check(dll().vitk_gemm_profile_read(v, C.byref(ms), C.byref(fl), C.byref(n), 0))
        if n.value:
            out.append(dict(name=dll().vitk_gemm_variant_name(v).decode(), ms=ms.value, flops=fl.value,
                            launches=n.value))
    if reset:
        check(dll().vitk_gemm_profile_read(0, None, None, None, 1))
    return out

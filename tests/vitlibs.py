"""ctypes loaders for the two CPU checkers used by the tests (never by the product):

* ``oracle()``  -> oracle/libvit_oracle.so  (this repo's C restatement of the reference path)
* ``ref()``     -> oracle/_ref/libvit_ref.so (the reference's own sources, compiled in place;
                   only exists where /root/reference was available at build time) or ``None``.

Signatures are given as short strings: p = pointer (numpy array or None), u = uint64,
f = float, i = int, l = long long, s = bytes path; the part after '>' is the return type
(v = void, f = float, u = uint64, i = int, P = void*).
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_T = {"p": C.c_void_p, "u": C.c_uint64, "f": C.c_float, "i": C.c_int, "l": C.c_longlong, "s": C.c_char_p,
      "P": C.c_void_p}
_R = {"v": None, "f": C.c_float, "u": C.c_uint64, "i": C.c_int, "P": C.c_void_p}


class Cfg(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("image_size", "patch_size", "in_channels", "num_classes", "embedding_dim",
                                          "num_heads", "num_layers", "mlp_hidden_dim")] + [("dropout_rate", C.c_float)]

    @classmethod
    def make(cls, image_size, patch_size, in_channels, num_classes, embedding_dim, num_heads, num_layers,
             mlp_hidden_dim, dropout_rate=0.0):
        return cls(image_size, patch_size, in_channels, num_classes, embedding_dim, num_heads, num_layers,
                   mlp_hidden_dim, dropout_rate)

    @property
    def tokens(self):
        return 1 + (self.image_size // self.patch_size) ** 2

    @property
    def patch_dim(self):
        return self.patch_size * self.patch_size * self.in_channels

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:-1]}
        d["dropout_rate"] = float(self.dropout_rate)
        return d


# the BASELINE configs (K=1000 assumed for the synthetic 224x224 ones) and the shipped checkpoint
CONFIGS = {
    "C1": dict(image_size=28, patch_size=14, in_channels=1, num_classes=10, embedding_dim=96, num_heads=8,
               num_layers=8, mlp_hidden_dim=384),
    "C2": dict(image_size=28, patch_size=7, in_channels=3, num_classes=8, embedding_dim=192, num_heads=6,
               num_layers=6, mlp_hidden_dim=768),
    "C3": dict(image_size=224, patch_size=16, in_channels=3, num_classes=1000, embedding_dim=192, num_heads=3,
               num_layers=12, mlp_hidden_dim=768),
    "C4": dict(image_size=224, patch_size=16, in_channels=3, num_classes=1000, embedding_dim=384, num_heads=6,
               num_layers=12, mlp_hidden_dim=1536),
    "C5": dict(image_size=224, patch_size=16, in_channels=3, num_classes=1000, embedding_dim=768, num_heads=12,
               num_layers=12, mlp_hidden_dim=3072),
    "CK": dict(image_size=28, patch_size=7, in_channels=1, num_classes=10, embedding_dim=64, num_heads=8,
               num_layers=6, mlp_hidden_dim=1024),
}


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "ctypes call needs a contiguous array"
        return a.ctypes.data
    if isinstance(a, C.Structure):
        return C.addressof(a)
    return a


class Lib:
    def __init__(self, path, sigs):
        self.path = path
        self.dll = C.CDLL(path)
        for name, sig in sigs.items():
            args, ret = sig.split(">")
            fn = getattr(self.dll, name)
            fn.argtypes = [_T[c] for c in args]
            fn.restype = _R[ret]
            setattr(self, name, self._wrap(fn, args))

    @staticmethod
    def _wrap(fn, args):
        def call(*a):
            assert len(a) == len(args), (fn.__name__, len(a), len(args))
            conv = [(_ptr(v) if c in "pP" else v) for v, c in zip(a, args)]
            return fn(*conv)
        return call


_ORACLE_SIGS = {
    "vo_matmul": "ppuuup>v",
    "vo_bmm": "ppuuuup>v",
    "vo_dense_forward": "puuuppp>v",
    "vo_dense_backward": "ppuuupppp>v",
    "vo_layernorm_forward": "puuppfppp>v",
    "vo_layernorm_backward": "ppppuuppp>v",
    "vo_gelu_forward": "pup>v",
    "vo_gelu_backward": "ppup>v",
    "vo_softmax_rows": "puup>v",
    "vo_softmax_rows_backward": "ppuup>v",
    "vo_mha": "uuppuupppp>v",
    "vo_ffn": "uuppuupppp>v",
    "vo_block": "uuuppuupppp>v",
    "vo_patch_gather": "puuuup>v",
    "vo_embeddings": "uuuuppuppp>v",
    "vo_cross_entropy": "ppuupup>f",
    "vo_adam_update": "ppppufffffff>v",
    "vo_cosine_warmup_lr": "lllf>f",
    "vo_argmax_rows": "puup>v",
    "vo_param_count": "p>u",
    "vo_param_tensors": "p>u",
    "vo_param_shape": "pup>u",
    "vo_forward": "pppupp>v",
    "vo_forward_backward": "ppppupuppp>f",
    "vo_session_create": "pffff>P",
    "vo_session_destroy": "P>v",
    "vo_session_set_params": "Pp>v",
    "vo_session_get_params": "Pp>v",
    "vo_session_set_class_weights": "Ppu>v",
    "vo_session_train_step": "Pppufpp>f",
}

_REF_SIGS = {
    "vref_create": "pfffff>P",
    "vref_destroy": "P>v",
    "vref_param_count": "P>u",
    "vref_param_tensors": "P>u",
    "vref_param_shape": "Pup>u",
    "vref_set_params": "Pp>v",
    "vref_get_params": "Pp>v",
    "vref_get_grads": "Pp>v",
    "vref_set_class_weights": "Ppu>v",
    "vref_save_weights": "Ps>i",
    "vref_load_weights": "Ps>i",
    "vref_save_config": "ps>i",
    "vref_load_config": "sp>i",
    "vref_forward": "Ppuip>v",
    "vref_forward_trace": "Ppupp>v",
    "vref_backward_trace": "Ppup>v",
    "vref_loss": "Pppu>f",
    "vref_train_step": "Pppufpp>f",
    "vref_forward_backward": "Pppup>f",
    "vref_matmul": "ppuuup>v",
    "vref_bmm": "ppuuuup>v",
    "vref_softmax3d": "puuup>v",
    "vref_softmax3d_backward": "ppuuup>v",
    "vref_dense": "uupppuppppp>v",
    "vref_layernorm": "ufpppuppppp>v",
    "vref_gelu": "puppp>v",
    "vref_mha": "uuppuupppp>v",
    "vref_ffn": "uuppuupppp>v",
    "vref_block": "uuuppuupppp>v",
    "vref_embeddings": "uuuuppupppp>v",
    "vref_cross_entropy": "ppuupup>f",
    "vref_adam": "uuffffuppp>v",
    "vref_cosine_warmup_lr": "lllf>f",
    "vref_epoch_order": "uip>v",
    "vref_argmax": "puup>v",
}

_cache = {}


def oracle():
    if "o" not in _cache:
        path = os.path.join(ROOT, "oracle", "libvit_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libvit_oracle.so missing: run `make -C oracle oracle` (or __graft_entry__.build())")
        _cache["o"] = Lib(path, _ORACLE_SIGS)
    return _cache["o"]


def ref():
    if "r" not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "libvit_ref.so")
        _cache["r"] = Lib(path, _REF_SIGS) if os.path.exists(path) else None
    return _cache["r"]


def f32(*shape):
    return np.zeros(shape, dtype=np.float32)


def rng_normal(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def onehot(labels, K):
    y = np.zeros((len(labels), K), dtype=np.float32)
    y[np.arange(len(labels)), labels] = 1.0
    return y


def param_shapes(cfg):
    """Shapes in getParameters() order (VisionTransformer.cpp:74-87 of the reference)."""
    D, H, K = cfg.embedding_dim, cfg.mlp_hidden_dim, cfg.num_classes
    shapes = [(cfg.patch_dim, D), (1, D), (1, 1, D), (1, cfg.tokens, D)]
    for _ in range(cfg.num_layers):
        shapes += [(1, D), (1, D)] + [(D, D), (1, D)] * 4 + [(1, D), (1, D), (D, H), (1, H), (H, D), (1, D)]
    shapes += [(1, D), (1, D), (D, K), (1, K)]
    return shapes


def init_params(cfg, seed, *, noise_bias=True):
    """Deterministic test weights: N(0, 0.02) matrices; LN gamma near 1; small non-zero biases so
    that every bias/beta path is exercised (the reference initialises them to 0/1 exactly)."""
    rng = np.random.default_rng(seed)
    out = []
    shapes = param_shapes(cfg)
    for i, s in enumerate(shapes):
        n = int(np.prod(s))
        is_gamma = False
        if 4 <= i < 4 + 16 * cfg.num_layers:
            is_gamma = (i - 4) % 16 in (0, 10)
        elif i == 4 + 16 * cfg.num_layers:
            is_gamma = True
        if is_gamma:
            v = 1.0 + (rng.standard_normal(n) * 0.05 if noise_bias else 0.0)
        elif s[0] == 1 and len(s) == 2:
            v = rng.standard_normal(n) * 0.02 if noise_bias else np.zeros(n)
        else:
            v = rng.standard_normal(n) * 0.02
        out.append(np.asarray(v, dtype=np.float32).reshape(-1))
    return np.concatenate(out)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_relerr(a, b, floor=None):
    """max |a-b| / max(|b|_inf-scale): elementwise error relative to the tensor's own scale."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = float(np.max(np.abs(b))) if floor is None else max(float(np.max(np.abs(b))), floor)
    return float(np.max(np.abs(a - b)) / max(scale, 1e-30))

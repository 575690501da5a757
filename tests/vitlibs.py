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
    "vo_block_masked": "uuuppuuppppp>v",
    "vo_forward_backward_masked": "ppppupuppppp>f",
    "vo_session_train_step_masked": "Pppufppp>f",
    "vo_patch_gather": "puuuup>v",
    "vo_embeddings": "uuuuppuppp>v",
    "vo_embeddings_input_grad": "uuuupupp>v",
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
    "vo_augment": "puuuuippp>v",
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
    "vref_block_dropout": "uuufppuuppppp>v",
    "vref_get_dropout_masks": "Pup>v",
    "vref_embeddings": "uuuuppupppp>v",
    "vref_cross_entropy": "ppuupup>f",
    "vref_adam": "uuffffuppp>v",
    "vref_cosine_warmup_lr": "lllf>f",
    "vref_epoch_order": "uip>v",
    "vref_argmax": "puup>v",
    "vref_augment": "puuuuipp>v",
    "vref_tensor_scenario": "ippppp>i",
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
            v = 1.0 + (rng.standard_normal(n) * 0.05 if noise_bias else np.zeros(n))
        elif s[0] == 1 and len(s) == 2:
            v = rng.standard_normal(n) * 0.02 if noise_bias else np.zeros(n)
        else:
            v = rng.standard_normal(n) * 0.02
        out.append(np.asarray(v, dtype=np.float32).reshape(-1))
    flat = np.concatenate(out)
    assert len(flat) == sum(int(np.prod(s)) for s in shapes)
    return flat


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


def elementwise_excess(a, b, tol, floor=None):
    """Largest |a-b| / (tol*|b| + tol*rms(b)) over the elements: <= 1 means every element is
    within tol of its own magnitude plus tol of the tensor's typical (root-mean-square) magnitude.
    Unlike max_relerr this does not let small elements hide behind the tensor's largest one."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    rms = float(np.sqrt(np.mean(b * b))) if b.size else 0.0
    if floor is not None:
        rms = max(rms, floor)
    bound = tol * np.abs(b) + tol * max(rms, 1e-30)
    return float(np.max(np.abs(a - b) / bound)) if b.size else 0.0


def assert_close(got, want, tol, what="", floor=None):
    """Both error bounds the parity tests use: the largest error relative to the tensor's largest
    magnitude (max_relerr) and the per-element bound of elementwise_excess."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    e = max_relerr(got, want, floor)
    assert e <= tol, f"{what}: max rel err {e:.3e} > {tol:.1e}"
    x = elementwise_excess(got, want, tol, floor)
    assert x <= 1.0, f"{what}: an element is off by {x:.2f} x (tol*|want| + tol*rms(want)), tol {tol:.1e}"


# ----------------------------------------------------------------------------- golden helpers
GOLDEN = os.path.join(ROOT, "tests", "golden")
TINY = dict(image_size=28, patch_size=14, in_channels=1, num_classes=10, embedding_dim=32, num_heads=4,
            num_layers=2, mlp_hidden_dim=64)


def tensor_digest(flat, shapes, nsamp=64):
    """Per-tensor (float64 sum, L2 norm, nsamp strided samples) of a flat parameter-ordered vector."""
    sums, norms, samples = [], [], []
    off = 0
    for shp in shapes:
        k = int(np.prod(shp))
        t = np.asarray(flat[off:off + k], dtype=np.float64)
        idx = np.linspace(0, k - 1, min(k, nsamp)).astype(np.int64)
        s = np.zeros(nsamp, dtype=np.float32)
        s[:len(idx)] = flat[off:off + k][idx]
        sums.append(t.sum()); norms.append(np.sqrt((t * t).sum())); samples.append(s)
        off += k
    return np.array(sums), np.array(norms), np.stack(samples)


def check_digest(flat, shapes, sums, norms, samples, tol, what=""):
    """Assert a flat vector matches a stored digest: samples and norm to `tol` relative to the
    tensor's scale, sum to tol * sqrt(k) * scale (a sum of k rounding errors)."""
    off = 0
    for i, shp in enumerate(shapes):
        k = int(np.prod(shp))
        t = np.asarray(flat[off:off + k], dtype=np.float64)
        idx = np.linspace(0, k - 1, min(k, samples.shape[1])).astype(np.int64)
        scale = max(float(np.max(np.abs(t))), float(np.max(np.abs(samples[i]))), 1e-30)
        got = t[idx]
        err = float(np.max(np.abs(got - samples[i][:len(idx)]))) / scale
        assert err <= tol, f"{what} tensor {i} {shp}: sample err {err:.3e} > {tol}"
        nerr = abs(np.sqrt((t * t).sum()) - norms[i]) / max(norms[i], 1e-30)
        assert nerr <= tol, f"{what} tensor {i} {shp}: norm err {nerr:.3e} > {tol}"
        serr = abs(t.sum() - sums[i]) / (scale * np.sqrt(k))
        assert serr <= tol, f"{what} tensor {i} {shp}: sum err {serr:.3e} > {tol}"
        off += k


def read_weights_file(path, shapes=None):
    """Parse the reference's .weights format (ModelUtils.cpp:28-51 of the reference): per tensor
    u64 rank, u64 dims[rank], f32 data[prod(dims)], native endian, no header."""
    raw = open(path, "rb").read()
    off, flat, got = 0, [], []
    while off < len(raw):
        rank = int(np.frombuffer(raw, np.uint64, 1, off)[0]); off += 8
        dims = tuple(int(d) for d in np.frombuffer(raw, np.uint64, rank, off)); off += 8 * rank
        n = int(np.prod(dims))
        flat.append(np.frombuffer(raw, np.float32, n, off)); off += 4 * n
        got.append(dims)
    if shapes is not None:
        assert [tuple(s) for s in shapes] == got, "checkpoint tensor shapes do not match the config"
    return np.concatenate(flat).astype(np.float32), got


def write_weights_file(path, flat, shapes):
    with open(path, "wb") as f:
        off = 0
        for shp in shapes:
            n = int(np.prod(shp))
            f.write(np.uint64(len(shp)).tobytes())
            f.write(np.asarray(shp, dtype=np.uint64).tobytes())
            f.write(np.asarray(flat[off:off + n], dtype=np.float32).tobytes())
            off += n


# ------------------------------------------------------------------ shared golden-case inputs
def op_case(name):
    """Seeded inputs of the op-level golden cases (tests/golden/ops.npz holds the reference's outputs)."""
    if name == "dense":
        rows, i, o = 24, 49, 40
        return dict(rows=rows, i=i, o=o, W=rng_normal(1, i, o, scale=0.05), b=rng_normal(2, 1, o, scale=0.05),
                    x=rng_normal(3, rows, i), dy=rng_normal(4, rows, o))
    if name == "ln":
        rows, D = 50, 96
        return dict(rows=rows, D=D, g=1 + rng_normal(5, 1, D, scale=0.1), b=rng_normal(6, 1, D, scale=0.1),
                    x=rng_normal(7, rows, D) * 2 + 0.3, dy=rng_normal(8, rows, D))
    if name == "gelu":
        return dict(n=2048, x=rng_normal(9, 2048) * 3, dy=rng_normal(10, 2048))
    if name == "softmax":
        return dict(b=6, n=17, x=rng_normal(11, 6, 17, 17) * 3, g=rng_normal(12, 6, 17, 17))
    if name in ("mha_a", "mha_b"):
        D, h, B, N = {"mha_a": (96, 8, 3, 5), "mha_b": (64, 8, 2, 17)}[name]
        n = 4 * (D * D + D)
        return dict(D=D, h=h, B=B, N=N, n=n, prm=rng_normal(13, n, scale=0.2), x=rng_normal(14, B, N, D),
                    dy=rng_normal(15, B, N, D))
    if name in ("ffn", "block"):
        D, H, h, B, N = 64, 160, 8, 3, 5
        n = D * H + H + H * D + D
        d = dict(D=D, H=H, h=h, B=B, N=N, x=rng_normal(17, B, N, D), dy=rng_normal(18, B, N, D))
        if name == "ffn":
            d.update(n=n, prm=rng_normal(16, n, scale=0.1))
        else:
            nb = 4 * D + 4 * (D * D + D) + n
            prm = rng_normal(19, nb, scale=0.1)
            prm[:D] += 1
            prm[2 * D + 4 * (D * D + D):3 * D + 4 * (D * D + D)] += 1
            d.update(n=nb, prm=prm)
        return d
    if name in ("emb_a", "emb_b", "emb_c"):
        img, P, Cc, D, B = {"emb_a": (28, 14, 1, 96, 4), "emb_b": (28, 7, 3, 48, 3), "emb_c": (32, 16, 3, 40, 2)}[name]
        N, pd = (img // P) ** 2 + 1, P * P * Cc
        n = pd * D + D + D + N * D
        return dict(img=img, P=P, C=Cc, D=D, B=B, N=N, pd=pd, n=n, prm=rng_normal(20, n, scale=0.05),
                    x=rng_normal(21, B, Cc, img, img), dy=rng_normal(22, B, N, D))
    if name == "ce":
        B, K = 37, 10
        w = np.abs(rng_normal(25, K)) + 0.1
        w /= w.sum()
        return dict(B=B, K=K, lg=rng_normal(23, B, K) * 2,
                    y=onehot(np.random.default_rng(24).integers(0, K, B), K), w=w.astype(np.float32))
    if name == "adam":
        r, c, steps = 7, 9, 5
        return dict(r=r, c=c, steps=steps, p=rng_normal(26, r, c), grads=rng_normal(27, steps, r, c),
                    lrs=np.array([0.0, 1e-4, 3e-4, 3e-4, 2e-4], dtype=np.float32))
    raise KeyError(name)


def model_case_inputs(cfgd, B, seed, steps):
    """Inputs of tests/golden/model_*.npz: params, first batch, per-step batches and learning rates."""
    cfg = Cfg.make(**cfgd)
    K = cfg.num_classes
    shape = (B, cfg.in_channels, cfg.image_size, cfg.image_size)
    d = dict(cfg=cfg, prm=init_params(cfg, seed), w=np.full(K, 1.0 / K, np.float32),
             x=rng_normal(seed + 1, *shape), y=onehot(np.random.default_rng(seed + 2).integers(0, K, B), K))
    d["xs"] = [rng_normal(seed + 10 + t, *shape) for t in range(steps)]
    d["ys"] = [onehot(np.random.default_rng(seed + 50 + t).integers(0, K, B), K) for t in range(steps)]
    return d


MODEL_CASES = {  # name -> (config, B, seed, steps, full_grads)
    "tiny_b4": (TINY, 4, 300, 3, True),
    "c1_b8": (CONFIGS["C1"], 8, 400, 3, False),
    "c2_b4": (CONFIGS["C2"], 4, 500, 2, False),
    "ck_b4": (CONFIGS["CK"], 4, 600, 2, False),
}


# ---- dropout > 0 (tests/golden/dropout.npz): the reference's masks are stored one bit per element
DROPOUT_RATE = 0.2  # the reference's training default (app/main.cpp:19)
DROPOUT_MODEL_CASE = (dict(CONFIGS["C1"], dropout_rate=DROPOUT_RATE), 8, 700, 2)  # config, B, seed, steps


def dropout_scale(rate):
    """Dropout::scale as the reference computes it in float (src/layers/Dropout.cpp:15)."""
    return np.float32(1.0) / (np.float32(1.0) - np.float32(rate))


def pack_masks(masks):
    return np.packbits(np.asarray(masks).reshape(-1) != 0)


def unpack_masks(bits, shape, rate):
    n = int(np.prod(shape))
    keep = np.unpackbits(bits)[:n].astype(np.float32)
    return (keep * dropout_scale(rate)).reshape(shape)


def dropout_block_case():
    D, H, h, B, N = 64, 160, 8, 3, 5
    n = 4 * D + 4 * (D * D + D) + D * H + H + H * D + D
    prm = rng_normal(30, n, scale=0.1)
    prm[:D] += 1
    prm[2 * D + 4 * (D * D + D):3 * D + 4 * (D * D + D)] += 1
    return dict(D=D, H=H, h=h, B=B, N=N, n=n, rate=DROPOUT_RATE, prm=prm, x=rng_normal(31, B, N, D),
                dy=rng_normal(32, B, N, D))


def loss_curve_inputs():
    """Inputs of tests/golden/loss_curve_c1_b32.npz (20 steps, C1, B=32, learnable template task)."""
    cfg = Cfg.make(**CONFIGS["C1"])
    B, steps, seed = 32, 20, 77
    templ = rng_normal(seed + 1, 10, 1, 28, 28)
    xs, ys = [], []
    for t in range(steps):
        lab = np.random.default_rng(seed + 100 + t).integers(0, 10, B)
        xs.append((templ[lab] + 0.5 * rng_normal(seed + 200 + t, B, 1, 28, 28)).astype(np.float32))
        ys.append(onehot(lab, 10))
    return dict(cfg=cfg, B=B, steps=steps, prm=init_params(cfg, seed, noise_bias=False),
                w=np.full(10, 0.1, np.float32), xs=xs, ys=ys)


# name -> (B, C, H, W, engine seed, DataAugmentation::Config fields in declaration order)
AUGMENT_CASES = {
    "defaults_mnist": (16, 1, 28, 28, 7, (0.5, 0.5, 0.5, 10.0, 0.1, 0.9, 1.1)),      # Trainer.cpp:157-161
    "all_stages_rgb": (6, 3, 28, 28, 11, (1.0, 1.0, 1.0, 30.0, 0.25, 0.7, 1.4)),
    "rectangular": (5, 2, 20, 33, 3, (0.8, 0.8, 0.8, 45.0, 0.3, 0.5, 2.0)),
    "rotate_only": (4, 1, 17, 17, 5, (1.0, 0.0, 0.0, 180.0, 0.1, 0.9, 1.1)),
    "none": (3, 1, 8, 8, 9, (0.0, 0.0, 0.0, 10.0, 0.1, 0.9, 1.1)),
}


def augment_case(name):
    B, Cc, H, W, seed, cfg = AUGMENT_CASES[name]
    return dict(B=B, C=Cc, H=H, W=W, seed=seed, cfg=np.array(cfg, np.float32), x=rng_normal(900 + seed, B, Cc, H, W))


# ---- Tensor-API scenarios (tests/tensor_surface.inc) -----------------------------------------
TENSOR_SCENARIOS = 30


def tensor_surface_inputs():
    return rng_normal(77, 4096), rng_normal(78, 4096)


def run_tensor_scenario(fn, sc):
    """fn(scenario, a, b, out, out_shape, out_rank) -> 0 of either build of the scenarios."""
    a, b = tensor_surface_inputs()
    out = np.zeros(8192, np.float32)
    shape = np.zeros(4, np.uint64)
    rank = np.zeros(1, np.uint64)
    rc = fn(sc, a.ctypes.data, b.ctypes.data, out.ctypes.data, shape.ctypes.data, rank.ctypes.data)
    assert rc == 0, (sc, rc)
    shp = tuple(int(v) for v in shape[:int(rank[0])])
    return out[:int(np.prod(shp))].reshape(shp).copy()


# ---- synthetic MNIST-format CSV for running the reference's applications -----------------------
def write_digit_csv(path, rows, seed, classes=10):
    """`label,pixel0..pixel783` with a header line, like the MNIST CSVs the reference's apps read
    (src/utils/DataReader.cpp:31-75): stroke images whose stroke pattern depends on the label, so a
    model can learn them; integer pixels 0..255."""
    rng = np.random.default_rng(seed)
    lines = ["label," + ",".join(f"pixel{i}" for i in range(784))]
    for _ in range(rows):
        lab = int(rng.integers(0, classes))
        img = rng.integers(0, 40, (28, 28))
        r0, c0 = 4 + 2 * (lab % 5), 4 + 4 * (lab // 5)
        dr, dc = (1, 0) if lab % 2 else (0, 1)
        for t in range(int(rng.integers(8, 14))):
            rr, cc = min(r0 + dr * t, 27), min(c0 + dc * t, 27)
            img[max(rr - 1, 0):rr + 2, max(cc - 1, 0):cc + 2] = rng.integers(200, 256)
        lines.append(str(lab) + "," + ",".join(str(int(v)) for v in img.reshape(-1)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


APP_TEST_ROWS, APP_TEST_SEED = 300, 2025


def golden(name):
    return np.load(os.path.join(GOLDEN, name))

"""ctypes binding of include/vit_host.h (lib/libvit_host.so): the C++ host layer -- the
reference-API Tensor / Layer / Loss / Optimizer / Trainer / ModelUtils classes -- behind opaque
handles.  Device pointers come from torch CUDA tensors."""
import ctypes as C
import os

import numpy as np

from . import vitk

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvit_host.so")


class HostError(RuntimeError):
    pass


class Config(C.Structure):
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

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:-1]}
        d["dropout_rate"] = float(self.dropout_rate)
        return d


_p, _u64, _i64, _i32, _f, _int = C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_float, C.c_int
_pf, _pu = C.POINTER(C.c_float), C.POINTER(C.c_uint64)

SIGNATURES = {
    "vith_last_error": (C.c_char_p, []),
    "vith_init": (_int, [_int]),
    "vith_set_stream": (_int, [_p]),
    "vith_synchronize": (_int, []),
    "vith_seed": (_int, [_u64]),
    "vith_set_fused_attention": (_int, [_int]),
    "vith_comm_unique_id": (_int, [_p]),
    "vith_comm_init": (_int, [_int, _int, _p]),
    "vith_comm_init_file": (_int, [_int, _int, C.c_char_p, _int]),
    "vith_comm_init_external": (_int, [_int, _int, _p, _p, _p, _int]),
    "vith_comm_exchange_id_file": (_int, [_int, _int, C.c_char_p, _int, _p]),
    "vith_comm_timing": (_int, [_int]),
    "vith_comm_exposed_ms": (_i64, [C.POINTER(C.c_double)]),
    "vith_last_fit_epoch_ms": (_u64, [C.POINTER(C.c_double), _u64]),
    "vith_comm_destroy": (_int, []),
    "vith_plan_buckets": (_i64, [_pu, _pu, _u64, _u64, _pu, _u64]),
    "vith_comm_rank": (_int, []),
    "vith_comm_world": (_int, []),
    "vith_comm_broadcast_params": (_int, [_p, _int]),
    "vith_comm_allreduce": (_int, [_p, _u64]),
    "vith_trainer_set_bucket_bytes": (_int, [_p, _u64]),
    "vith_trainer_num_buckets": (_u64, [_p]),
    "vith_dropout_state": (_int, [_pu, _pu]),
    "vith_set_dropout_state": (_int, [_u64, _u64]),
    "vith_dropout_replay": (_int, [_p, _u64]),
    "vith_set_first_token_tail": (_int, [_int]),
    "vith_device_bytes_in_use": (_u64, []),
    "vith_model_create": (_p, [C.POINTER(Config), _int, _int]),
    "vith_model_destroy": (None, [_p]),
    "vith_param_count": (_u64, [_p]),
    "vith_param_tensors": (_u64, [_p]),
    "vith_param_shape": (_u64, [_p, _u64, _pu]),
    "vith_get_params": (_int, [_p, _p]),
    "vith_set_params": (_int, [_p, _p]),
    "vith_get_grads": (_int, [_p, _p]),
    "vith_params_device": (_p, [_p]),
    "vith_grads_device": (_p, [_p]),
    "vith_save_weights": (_int, [_p, C.c_char_p]),
    "vith_load_weights": (_int, [_p, C.c_char_p]),
    "vith_save_config": (_int, [C.POINTER(Config), C.c_char_p]),
    "vith_load_config": (_int, [C.c_char_p, C.POINTER(Config)]),
    "vith_forward": (_int, [_p, _p, _u64, _int, _p]),
    "vith_forward_trace": (_int, [_p, _p, _u64, _p, _p]),
    "vith_backward": (_int, [_p, _p, _u64, _p]),
    "vith_backward_image": (_int, [_p, _p, _u64, _p]),
    "vith_predict": (_int, [_p, _p, _u64, _p]),
    "vith_trainer_create": (_p, [_p, _f, _f, _f, _f, _p, _u64]),
    "vith_trainer_destroy": (None, [_p]),
    "vith_train_step": (_int, [_p, _p, _p, _u64, _f, _pf, _pf]),
    "vith_step_forward": (_int, [_p, _p, _p, _u64, _f]),
    "vith_num_stages": (_u64, [_p]),
    "vith_step_backward_stage": (_int, [_p, _u64]),
    "vith_fit": (_int, [_p, _p, _p, _u64, _p, _p, _u64, _int, _u64, C.c_float, C.c_float, C.c_float, _p]),
    "vith_fit_augmented": (_int, [_p, _p, _p, _u64, _p, _p, _u64, _int, _u64, C.c_float, C.c_float, C.c_float, _p,
                                  C.c_uint32, _p]),
    "vith_fit_reference_defaults": (_int, [_p, _p, _p, _u64, _p, _p, _u64, _int, _u64, C.c_float, C.c_float, C.c_float,
                                           _p]),
    "vith_next_seed": (_u64, []),
    "vith_fit_with_options": (_int, [_p, _p, _p, _u64, _p, _p, _u64, _int, _u64, C.c_float, C.c_float, C.c_float, _p,
                                     _p]),
    "vith_augment": (_int, [_p, C.c_uint32, _p, _u64, _u64, _u64, _u64, _p]),
    "vith_trainer_save_state": (_int, [_p, C.c_char_p]),
    "vith_trainer_load_state": (_int, [_p, C.c_char_p]),
    "vith_stage_grad_range": (_int, [_p, _u64, _pu, _pu]),
    "vith_step_update": (_int, [_p, _f, _f]),
    "vith_step_result": (_int, [_p, _pf, _pf]),
    "vith_eval_batch": (_int, [_p, _p, _p, _u64, _pf, _pf]),
    "vith_cosine_warmup_lr": (_f, [_i64, _i64, _i64, _f]),
    "vith_epoch_order": (_int, [_u64, _i32, _p]),
    "vith_layer_create": (_p, [_int, _u64, _u64, _u64, _u64]),
    "vith_layer_create_dropout": (_p, [_int, _u64, _u64, _u64, _u64, _f]),
    "vith_layer_destroy": (None, [_p]),
    "vith_layer_param_count": (_u64, [_p]),
    "vith_layer_set_params": (_int, [_p, _p]),
    "vith_layer_get_grads": (_int, [_p, _p]),
    "vith_layer_forward": (_int, [_p, _p, _pu, _u64, _int, _p, _u64]),
    "vith_layer_backward": (_int, [_p, _p, _pu, _u64, _p, _u64]),
}

_dll = None


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise HostError(f"{LIB_PATH} is missing: build it with `make -C visiontransformer_amd` "
                            "(or __graft_entry__.build()); there is no CPU fallback")
        vitk.dll()  # libvitk.so first (libvit_host.so links against it)
        d = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(d, name)
            fn.restype, fn.argtypes = res, args
        _dll = d
    return _dll


def _check(rc):
    if rc != 0:
        raise HostError(dll().vith_last_error().decode(errors="replace"))


def _np(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data


def init(device=0):
    _check(dll().vith_init(device))


def use_torch_stream():
    """Enqueue all host-layer work on torch's current CUDA stream."""
    import torch
    _check(dll().vith_set_stream(torch.cuda.current_stream().cuda_stream))


def set_fused_attention(on):
    """Attention path of models / layers created afterwards (see include/vit_host.h)."""
    _check(dll().vith_set_fused_attention(int(bool(on))))


def set_first_token_tail(on):
    """Last block + final norm on the CLS token only (default on; see include/vit_host.h)."""
    _check(dll().vith_set_first_token_tail(int(bool(on))))


def seed(s):
    _check(dll().vith_seed(s))


# ---- data parallelism (include/vit_host.h: vith_comm_*) ----------------------------------------
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p)
BROADCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p)
_transport_keepalive = []


def comm_unique_id():
    """The 128-byte id rank 0 creates for a new RCCL communicator."""
    buf = C.create_string_buffer(128)
    _check(dll().vith_comm_unique_id(buf))
    return buf.raw


def comm_init(rank, world, unique_id):
    """Join the RCCL communicator (collective over all ranks)."""
    assert len(unique_id) == 128
    _check(dll().vith_comm_init(rank, world, unique_id))


def comm_init_file(rank, world, path, timeout_s=120):
    _check(dll().vith_comm_init_file(rank, world, os.fsencode(path), timeout_s))


def comm_exchange_id_file(rank, world, path, unique_id=None, timeout_s=120):
    """The id exchange of comm_init_file alone (no RCCL, no GPU): rank 0 passes the id, every rank
    returns it."""
    buf = C.create_string_buffer(unique_id if rank == 0 else b"", 128)
    _check(dll().vith_comm_exchange_id_file(rank, world, os.fsencode(path), timeout_s, buf))
    return buf.raw


def comm_timing(on):
    _check(dll().vith_comm_timing(int(bool(on))))


def comm_exposed_ms():
    """(sum of the per-step exposed exchange times in ms, number of steps) since the last call."""
    total = C.c_double()
    n = dll().vith_comm_exposed_ms(C.byref(total))
    if n < 0:
        raise HostError(dll().vith_last_error().decode(errors="replace"))
    return total.value, int(n)


def last_fit_epoch_ms():
    n = dll().vith_last_fit_epoch_ms(None, 0)
    out = (C.c_double * max(int(n), 1))()
    dll().vith_last_fit_epoch_ms(out, n)
    return [out[i] for i in range(int(n))]


def comm_init_external(rank, world, allreduce, broadcast, stream_ordered):
    """allreduce(dev_ptr, count, stream) / broadcast(dev_ptr, count, root, stream): Python callables
    working in place on device memory; exceptions they raise become a transport failure."""
    errors = []

    def ar(dev, count, stream, _user):
        try:
            allreduce(dev, count, stream)
            return 0
        except Exception as e:  # noqa: BLE001 - must not unwind through the C++ caller
            errors.append(e)
            return 1

    def bc(dev, count, root, stream, _user):
        try:
            broadcast(dev, count, root, stream)
            return 0
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            return 1

    fns = (ALLREDUCE_FN(ar), BROADCAST_FN(bc), errors)
    _transport_keepalive.append(fns)
    _check(dll().vith_comm_init_external(rank, world, C.cast(fns[0], C.c_void_p), C.cast(fns[1], C.c_void_p), None,
                                         int(stream_ordered)))
    return errors


def plan_buckets(ranges, bucket_bytes):
    """vit::GradientReducer::planBuckets: [(last_stage, offset, count), ...] for stage slices
    ranges[s] = (offset, count)."""
    n = len(ranges)
    off = (C.c_uint64 * n)(*[r[0] for r in ranges])
    cnt = (C.c_uint64 * n)(*[r[1] for r in ranges])
    out = (C.c_uint64 * (3 * max(n, 1)))()
    k = dll().vith_plan_buckets(off, cnt, n, bucket_bytes, out, max(n, 1))
    if k < 0:
        raise HostError(dll().vith_last_error().decode(errors="replace"))
    return [(int(out[3 * i]), int(out[3 * i + 1]), int(out[3 * i + 2])) for i in range(k)]


def comm_destroy():
    _check(dll().vith_comm_destroy())
    _transport_keepalive.clear()


def comm_rank():
    return dll().vith_comm_rank()


def comm_world():
    return dll().vith_comm_world()


def comm_allreduce(t):
    """In-place sum of a float32 CUDA tensor over the ranks, on the compute stream."""
    _check(dll().vith_comm_allreduce(t.data_ptr(), t.numel()))


def dropout_state():
    """(seed, offset) the next training-mode Dropout forward will draw its mask from."""
    s, o = C.c_uint64(), C.c_uint64()
    _check(dll().vith_dropout_state(C.byref(s), C.byref(o)))
    return s.value, o.value


def set_dropout_state(seed_, offset):
    _check(dll().vith_set_dropout_state(seed_, offset))


def dropout_replay(masks):
    """Dropout layers take their factors from `masks` (a CUDA tensor, call order) until dropout_replay(None)."""
    if masks is None:
        _check(dll().vith_dropout_replay(None, 0))
    else:
        _check(dll().vith_dropout_replay(masks.data_ptr(), masks.numel()))


def dropout_masks(rate, seed_, offset, counts):
    """The masks that consecutive Dropout forwards over counts[0], counts[1], ... elements draw from
    stream position (seed_, offset): a list of CUDA tensors holding 0 or 1/(1-rate)."""
    import torch
    out = []
    for n in counts:
        ones = torch.ones(n, dtype=torch.float32, device="cuda")
        y, m = torch.empty_like(ones), torch.empty_like(ones)
        vitk.call("vitk_dropout_fwd", ones, n, rate, seed_, offset, y, m, vitk.stream())
        out.append(m)
        offset += (n + 3) // 4
    return out


def next_seed():
    return dll().vith_next_seed()


def synchronize():
    _check(dll().vith_synchronize())


def cosine_warmup_lr(step, warmup, total, lr_max):
    return dll().vith_cosine_warmup_lr(step, warmup, total, lr_max)


class FitOptions(C.Structure):
    _fields_ = [("best_weights_path", C.c_char_p), ("state_path", C.c_char_p), ("start_epoch", C.c_int32),
                ("resume", C.c_int32), ("stop_epoch", C.c_int32)]


class AugmentConfig(C.Structure):
    """DataAugmentation::Config (include/utils/DataAugmentation.hpp:11-19 of the reference)."""
    _fields_ = [(n, C.c_float) for n in ("rotation_prob", "translate_prob", "zoom_prob", "rotation_factor",
                                         "translate_factor", "zoom_min", "zoom_max")]

    @classmethod
    def make(cls, rotation_prob=0.5, translate_prob=0.5, zoom_prob=0.5, rotation_factor=10.0, translate_factor=0.1,
             zoom_min=0.9, zoom_max=1.1):
        return cls(rotation_prob, translate_prob, zoom_prob, rotation_factor, translate_factor, zoom_min, zoom_max)


def augment(cfg, seed, x, out=None):
    """DataAugmentation(cfg).apply(x) with the engine seeded by `seed`; x is a {B,C,H,W} CUDA tensor."""
    import torch
    if out is None:
        out = torch.empty_like(x)
    B, Cc, Hh, W = x.shape
    _check(dll().vith_augment(C.byref(cfg), seed, x.data_ptr(), B, Cc, Hh, W, out.data_ptr()))
    return out


def epoch_order(n, epoch):
    out = np.zeros(n, dtype=np.uint64)
    _check(dll().vith_epoch_order(n, epoch, out.ctypes.data))
    return out


def save_config(cfg, path):
    _check(dll().vith_save_config(C.byref(cfg), os.fsencode(path)))


def load_config(path):
    cfg = Config()
    _check(dll().vith_load_config(os.fsencode(path), C.byref(cfg)))
    return cfg


class Model:
    """VisionTransformer handle."""

    def __init__(self, cfg, patch_mode=vitk.PATCH_REFERENCE, dk_mode=vitk.DK_REFERENCE):
        self.cfg = cfg
        self.h = dll().vith_model_create(C.byref(cfg), patch_mode, dk_mode)
        if not self.h:
            raise HostError(dll().vith_last_error().decode(errors="replace"))
        self.n = dll().vith_param_count(self.h)

    def close(self):
        if self.h:
            dll().vith_model_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def param_shapes(self):
        out = []
        dims = (C.c_uint64 * 4)()
        for i in range(dll().vith_param_tensors(self.h)):
            r = dll().vith_param_shape(self.h, i, dims)
            out.append(tuple(int(dims[k]) for k in range(r)))
        return out

    def set_params(self, flat):
        a, p = _np(flat)
        assert a.size == self.n, (a.size, self.n)
        _check(dll().vith_set_params(self.h, p))

    def get_params(self):
        out = np.zeros(self.n, dtype=np.float32)
        _check(dll().vith_get_params(self.h, out.ctypes.data))
        return out

    def get_grads(self):
        out = np.zeros(self.n, dtype=np.float32)
        _check(dll().vith_get_grads(self.h, out.ctypes.data))
        return out

    def broadcast_params(self, root=0):
        """Data parallelism: every replica takes rank `root`'s parameters."""
        _check(dll().vith_comm_broadcast_params(self.h, root))

    def params_device(self):
        return dll().vith_params_device(self.h)

    def grads_device(self):
        return dll().vith_grads_device(self.h)

    def fit(self, x, y, x_test, y_test, epochs, batch_size, lr_init=3e-4, weight_decay=1e-4, warmup_frac=0.1,
            augment=None, augment_seed=0):
        """Trainer::train on host numpy arrays; returns [epochs][4] = train loss/acc, test loss/acc.
        augment: an AugmentConfig to augment every training batch (engine seed augment_seed + epoch)."""
        x, y, x_test, y_test = (np.ascontiguousarray(a, np.float32) for a in (x, y, x_test, y_test))
        hist = np.zeros((epochs, 4), np.float32)
        _check(dll().vith_fit_augmented(self.h, x.ctypes.data, y.ctypes.data, x.shape[0], x_test.ctypes.data,
                                        y_test.ctypes.data, x_test.shape[0], epochs, batch_size, lr_init, weight_decay,
                                        warmup_frac, C.byref(augment) if augment is not None else None, augment_seed,
                                        hist.ctypes.data))
        return hist

    def fit_reference_defaults(self, x, y, x_test, y_test, epochs, batch_size, lr_init=3e-4, weight_decay=1e-4,
                               warmup_frac=0.1):
        """Trainer::train with a default-constructed TrainerConfig (augmentation on, as in the reference)."""
        x, y, x_test, y_test = (np.ascontiguousarray(a, np.float32) for a in (x, y, x_test, y_test))
        hist = np.zeros((epochs, 4), np.float32)
        _check(dll().vith_fit_reference_defaults(self.h, x.ctypes.data, y.ctypes.data, x.shape[0], x_test.ctypes.data,
                                                 y_test.ctypes.data, x_test.shape[0], epochs, batch_size, lr_init,
                                                 weight_decay, warmup_frac, hist.ctypes.data))
        return hist

    def fit_with_options(self, x, y, x_test, y_test, epochs, batch_size, lr_init=3e-4, weight_decay=1e-4,
                         warmup_frac=0.1, best_weights_path=None, state_path=None, start_epoch=0, resume=False,
                         stop_epoch=0):
        """vith_fit with TrainerConfig's checkpoint extensions (see include/vit_host.h)."""
        x, y, x_test, y_test = (np.ascontiguousarray(a, np.float32) for a in (x, y, x_test, y_test))
        opt = FitOptions(os.fsencode(best_weights_path) if best_weights_path else None,
                         os.fsencode(state_path) if state_path else None, start_epoch, int(resume), stop_epoch)
        hist = np.zeros((epochs, 4), np.float32)
        _check(dll().vith_fit_with_options(self.h, x.ctypes.data, y.ctypes.data, x.shape[0], x_test.ctypes.data,
                                           y_test.ctypes.data, x_test.shape[0], epochs, batch_size, lr_init,
                                           weight_decay, warmup_frac, C.byref(opt), hist.ctypes.data))
        return hist

    def save_weights(self, path):
        _check(dll().vith_save_weights(self.h, os.fsencode(path)))

    def load_weights(self, path):
        _check(dll().vith_load_weights(self.h, os.fsencode(path)))

    def forward(self, x, training=False, out=None):
        import torch
        B = x.shape[0]
        if out is None:
            out = torch.empty(B, self.cfg.num_classes, dtype=torch.float32, device=x.device)
        _check(dll().vith_forward(self.h, x.data_ptr(), B, int(training), out.data_ptr()))
        return out

    def forward_trace(self, x):
        import torch
        B, c = x.shape[0], self.cfg
        acts = torch.empty(c.num_layers + 2, B, c.tokens, c.embedding_dim, dtype=torch.float32, device=x.device)
        logits = torch.empty(B, c.num_classes, dtype=torch.float32, device=x.device)
        _check(dll().vith_forward_trace(self.h, x.data_ptr(), B, acts.data_ptr(), logits.data_ptr()))
        return acts, logits

    def backward(self, dlogits, trace=False):
        import torch
        B, c = dlogits.shape[0], self.cfg
        dacts = None
        if trace:
            dacts = torch.empty(c.num_layers + 1, B, c.tokens, c.embedding_dim, dtype=torch.float32,
                                device=dlogits.device)
        _check(dll().vith_backward(self.h, dlogits.data_ptr(), B, dacts.data_ptr() if trace else None))
        return dacts

    def backward_image(self, dlogits):
        """VisionTransformer::backward's return value: the gradient w.r.t. the input images."""
        import torch
        B, c = dlogits.shape[0], self.cfg
        out = torch.empty(B, c.in_channels, c.image_size, c.image_size, dtype=torch.float32, device=dlogits.device)
        _check(dll().vith_backward_image(self.h, dlogits.data_ptr(), B, out.data_ptr()))
        return out

    def predict(self, x):
        import torch
        out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
        _check(dll().vith_predict(self.h, x.data_ptr(), x.shape[0], out.data_ptr()))
        return out


class Trainer:
    """CrossEntropy + Adam(L2) around one Model: the per-batch body of the reference's Trainer."""

    def __init__(self, model, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4, class_weights=None):
        self.model = model
        w, n = None, 0
        if class_weights is not None:
            self._w, w = _np(class_weights)
            n = self._w.size
        self.h = dll().vith_trainer_create(model.h, beta1, beta2, eps, weight_decay, w, n)
        if not self.h:
            raise HostError(dll().vith_last_error().decode(errors="replace"))
        self.num_stages = dll().vith_num_stages(self.h)

    def close(self):
        if self.h:
            dll().vith_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def train_step(self, x, y, lr, want_result=True):
        loss, ok = C.c_float(), C.c_float()
        _check(dll().vith_train_step(self.h, x.data_ptr(), y.data_ptr(), x.shape[0], lr,
                                     C.byref(loss) if want_result else None, C.byref(ok) if want_result else None))
        return (loss.value, ok.value) if want_result else None

    def step_forward(self, x, y, grad_batch=0.0):
        _check(dll().vith_step_forward(self.h, x.data_ptr(), y.data_ptr(), x.shape[0], grad_batch))

    def backward_stage(self, s):
        _check(dll().vith_step_backward_stage(self.h, s))

    def set_bucket_bytes(self, n):
        _check(dll().vith_trainer_set_bucket_bytes(self.h, int(n)))

    def num_buckets(self):
        return dll().vith_trainer_num_buckets(self.h)

    def save_state(self, path):
        _check(dll().vith_trainer_save_state(self.h, os.fsencode(path)))

    def load_state(self, path):
        _check(dll().vith_trainer_load_state(self.h, os.fsencode(path)))

    def stage_grad_range(self, s):
        o, c = C.c_uint64(), C.c_uint64()
        _check(dll().vith_stage_grad_range(self.h, s, C.byref(o), C.byref(c)))
        return o.value, c.value

    def update(self, lr, grad_scale=1.0):
        _check(dll().vith_step_update(self.h, lr, grad_scale))

    def result(self):
        loss, ok = C.c_float(), C.c_float()
        _check(dll().vith_step_result(self.h, C.byref(loss), C.byref(ok)))
        return loss.value, ok.value

    def eval_batch(self, x, y):
        loss, ok = C.c_float(), C.c_float()
        _check(dll().vith_eval_batch(self.h, x.data_ptr(), y.data_ptr(), x.shape[0], C.byref(loss), C.byref(ok)))
        return loss.value, ok.value


LAYER_KINDS = {"dense": 0, "layernorm": 1, "gelu": 2, "mha": 3, "ffn": 4, "block": 5, "embeddings": 6}


class Layer:
    """One reference Layer subclass (host-buffer in/out; for op-level parity tests)."""

    def __init__(self, kind, a=0, b=0, c=0, d=0, dropout_rate=0.0):
        self.h = dll().vith_layer_create_dropout(LAYER_KINDS[kind], a, b, c, d, dropout_rate)
        if not self.h:
            raise HostError(dll().vith_last_error().decode(errors="replace"))
        self.n = dll().vith_layer_param_count(self.h)

    def close(self):
        if self.h:
            dll().vith_layer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_params(self, flat):
        a, p = _np(flat)
        assert a.size == self.n, (a.size, self.n)
        _check(dll().vith_layer_set_params(self.h, p))

    def get_grads(self):
        out = np.zeros(self.n, dtype=np.float32)
        _check(dll().vith_layer_get_grads(self.h, out.ctypes.data))
        return out

    def forward(self, x, out_shape, training=True):
        a, p = _np(x)
        shp = (C.c_uint64 * a.ndim)(*a.shape)
        out = np.zeros(out_shape, dtype=np.float32)
        _check(dll().vith_layer_forward(self.h, p, shp, a.ndim, int(training), out.ctypes.data, out.size))
        return out

    def backward(self, dy, dx_shape=None):
        a, p = _np(dy)
        shp = (C.c_uint64 * a.ndim)(*a.shape)
        if dx_shape is None:
            _check(dll().vith_layer_backward(self.h, p, shp, a.ndim, None, 0))
            return None
        out = np.zeros(dx_shape, dtype=np.float32)
        _check(dll().vith_layer_backward(self.h, p, shp, a.ndim, out.ctypes.data, out.size))
        return out

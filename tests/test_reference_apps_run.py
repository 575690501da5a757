"""The reference's own applications and input pipeline, RUN (not just linked).

* app/test.cpp (evaluation: accuracy, per-class precision / recall / F1, confusion matrix), built
  unchanged against the host layer (tests/native/app_test_host), on a synthetic MNIST-format CSV
  and the shipped vit_mnist_4_8 checkpoint: its stdout must equal, byte for byte, what the same
  source built against the reference's own sources printed (tests/golden/app_test_stdout.txt,
  tests/golden/make_golden.py apps).
* app/main.cpp (training, dropout 0.2, augmentation, class weights, checkpoint writing) and
  app/tl.cpp (fine-tuning from a checkpoint): random by construction in the reference
  (random_device), so these are checked for what does not depend on the draw -- they run to the
  end, print every epoch, learn, and write the files the reference writes.
* load_csv_data_train_val / compute_class_weights (src/utils/DataReader.cpp:31-194) through probes
  written once and compiled against both libraries (tests/data_reader.inc).
"""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import vitlibs as V

NATIVE = os.path.join(V.ROOT, "tests", "native")


def reader_probe(dll, prefix, csv, n, classes=10, sample=1.0, train=0.8, val=0.2):
    fn = getattr(dll, prefix + "_data_reader")
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_uint64] + [C.c_void_p] * 4
    labels, sums = np.full(n, -7, np.int32), np.zeros(n, np.float32)
    counts, weights = np.zeros(2, np.uint64), np.zeros(classes, np.float32)
    assert fn(os.fsencode(csv), sample, train, val, classes, labels.ctypes.data, sums.ctypes.data, counts.ctypes.data,
              weights.ctypes.data) == 0
    return labels, sums, [int(c) for c in counts], weights


def expected_from_csv(csv):
    raw = np.loadtxt(csv, delimiter=",", skiprows=1, dtype=np.float32)
    labels = raw[:, 0].astype(np.int32)
    px = (raw[:, 1:] / np.float32(255.0) - np.float32(0.1307)) / np.float32(0.3081)  # DataReader.cpp:71-72
    sums = np.zeros(len(px), np.float32)
    for j in range(px.shape[1]):  # float accumulation in pixel order, like the probe
        sums += px[:, j]
    return labels, sums


def class_weights_numpy(labels, classes):
    counts = np.bincount(labels, minlength=classes).astype(np.float32)
    w = np.float32(len(labels)) / (np.float32(classes) * counts)
    s = np.float32(0)
    for v in w:
        s = np.float32(s + v)
    return (w / s).astype(np.float32)


def check_reader(dll, prefix, tmp_path):
    csv = str(tmp_path / "train.csv")
    n = 203
    V.write_digit_csv(csv, n, 11)
    want_labels, want_sums = expected_from_csv(csv)
    labels, sums, (ntr, nva), weights = reader_probe(dll, prefix, csv, n)
    assert (ntr, nva) == (int(n * np.float32(0.8)), int(n * np.float32(0.2)))
    k = ntr + nva
    # the loader shuffles with random_device: compare as multisets of (label, checksum)
    got = sorted(zip(labels[:k].tolist(), sums[:k].tolist()))
    full = sorted(zip(want_labels.tolist(), want_sums.tolist()))
    assert len(set(got)) == k and set(got) <= set(full)
    np.testing.assert_allclose(weights, class_weights_numpy(labels[:ntr], 10), rtol=1e-6)
    # a second load draws another order (random_device), same content
    labels2, sums2, _, _ = reader_probe(dll, prefix, csv, n)
    assert labels2[:k].tolist() != labels[:k].tolist()
    # sample fraction
    _, _, (ntr3, nva3), _ = reader_probe(dll, prefix, csv, n, sample=0.5)
    assert (ntr3, nva3) == (int(int(n * 0.5) * np.float32(0.8)), int(int(n * 0.5) * np.float32(0.2)))
    # class weights on fixed labels, unbalanced
    fn = getattr(dll, prefix + "_class_weights")
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lab = np.array([0] * 50 + [1] * 5 + [2] * 20 + [3] * 25)
    y = V.onehot(lab, 4)
    w = np.zeros(4, np.float32)
    assert fn(y.ctypes.data, len(lab), 4, w.ctypes.data) == 0
    return w, class_weights_numpy(lab, 4)


@pytest.mark.skipif(V.ref() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_reference_data_reader_matches_the_numpy_restatement(tmp_path):
    w, want = check_reader(V.ref().dll, "vref", tmp_path)
    np.testing.assert_array_equal(w, want)
    np.save(tmp_path / "w.npy", w)


@pytest.mark.gpu
def test_host_layer_data_reader_matches_the_reference(tmp_path):
    from visiontransformer_amd import host
    host.init(0)
    dll = C.CDLL(os.path.join(NATIVE, "libtensor_surface.so"))
    w, want = check_reader(dll, "vith_test", tmp_path)
    np.testing.assert_array_equal(w, want)  # = what the reference gives (CPU test above)


def stage(tmp_path, files):
    os.makedirs(tmp_path / "data", exist_ok=True)
    os.makedirs(tmp_path / "models", exist_ok=True)
    for dst, src in files.items():
        shutil.copyfile(src, tmp_path / "models" / dst)


def run_app(name, cwd, timeout=110):
    exe = os.path.join(NATIVE, name)
    assert os.path.exists(exe), "built by `make -C tests/native` where the reference tree exists"
    lib = os.path.join(os.path.dirname(os.path.dirname(NATIVE)), "visiontransformer_amd", "lib", "libvit_host.so")
    # the applications hold the classes' layouts from the headers they were compiled against
    assert os.path.getmtime(exe) >= os.path.getmtime(lib), f"{name} is older than libvit_host.so: rebuild with `make -C tests/native`"
    r = subprocess.run([exe], cwd=cwd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


@pytest.mark.gpu
def test_reference_evaluation_app_prints_what_the_reference_prints(tmp_path):
    stage(tmp_path, {f"vit_mnist_4_8.{e}": os.path.join(V.GOLDEN, f"vit_mnist_4_8.{e}") for e in ("json", "weights")})
    V.write_digit_csv(str(tmp_path / "data" / "mnist_test.csv"), V.APP_TEST_ROWS, V.APP_TEST_SEED)
    out = run_app("app_test_host", tmp_path)
    want = open(os.path.join(V.GOLDEN, "app_test_stdout.txt")).read()
    assert out == want


EPOCH = re.compile(r"--- Epoch (\d+)/(\d+).*?Train Loss: ([0-9.]+) \| Train Acc: ([0-9.]+).*?Test Loss: ([0-9.]+) \| "
                   r"Test Acc: ([0-9.]+)")


@pytest.mark.gpu
def test_reference_training_and_fine_tuning_apps_run_to_the_end(tmp_path):
    stage(tmp_path, {})
    V.write_digit_csv(str(tmp_path / "data" / "mnist_train.csv"), 1280, 7)
    out = run_app("app_main_host", tmp_path)
    assert "--- Cargando data/mnist_train.csv  (fracción: 50%, μ=0.1307, σ=0.3081)" in out
    assert "Carga completa. 640 muestras cargadas." in out
    assert all(f"Clase {c}: " in out for c in range(10))  # print_classweights
    ep = [tuple(float(v) for v in m.groups()) for m in EPOCH.finditer(out)]
    assert [int(e[0]) for e in ep] == list(range(1, 31)) and all(e[1] == 30 for e in ep)
    assert ep[-1][2] < 0.6 * ep[0][2], (ep[0], ep[-1])   # it learns (dropout 0.2, augmentation on)
    assert ep[-1][5] > 0.5                               # and generalises to the validation split
    assert "¡Entrenamiento completado!" in out and "Proceso finalizado." in out
    rows = open(tmp_path / "vit_results.csv").read().strip().split("\n")
    assert rows[0] == "epoch,epochs_total,train_loss,train_acc,test_loss,test_acc" and len(rows) == 31
    assert rows[-1].startswith("30,30,")
    # the checkpoint it wrote is a C1-shaped model in the reference's format
    cfg = V.Cfg.make(**dict(V.CONFIGS["C1"], dropout_rate=0.2))
    flat, _ = V.read_weights_file(str(tmp_path / "models" / "vit_mnist_test.weights"), V.param_shapes(cfg))
    assert np.isfinite(flat).all()
    import json
    assert json.load(open(tmp_path / "models" / "vit_mnist_test.json")) == pytest.approx(cfg.as_dict())
    # fine-tuning app: starts from a checkpoint under the name it expects
    for e in ("json", "weights"):
        shutil.copyfile(tmp_path / "models" / f"vit_mnist_test.{e}", tmp_path / "models" / f"vit_mnist_test_64emb_30ep.{e}")
    out2 = run_app("app_tl_host", tmp_path)
    ep2 = [tuple(float(v) for v in m.groups()) for m in EPOCH.finditer(out2)]
    assert [int(e[0]) for e in ep2] == list(range(1, 11))
    # it continued from the trained weights: first-epoch loss well under ln(10) = 2.30 of a fresh model
    # (this app passes no class weights, so its loss is not scaled by ~1/10 like the training app's)
    assert ep2[0][2] < 0.6 * np.log(10.0), ep2[0]
    assert "Fine-tuning finalizado." in out2
    assert os.path.exists(tmp_path / "models" / "vit_mnist_test_64emb_30ep_finetuned.weights")

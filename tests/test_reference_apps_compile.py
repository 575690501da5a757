"""The drop-in claim, checked literally: the reference's own applications (app/main.cpp = the
training entry point, app/test.cpp = evaluation, app/tl.cpp = fine-tuning) compile and link,
UNCHANGED and in place, against the host layer's headers and libraries.  Needs /root/reference
(skipped elsewhere) and no GPU (nothing is executed)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "app")), reason="reference tree not present")


@pytest.mark.parametrize("app", ["main", "test", "tl"])
def test_reference_app_builds_against_host_layer(app, tmp_path):
    lib = os.path.join(ROOT, "visiontransformer_amd", "lib")
    assert os.path.exists(os.path.join(lib, "libvit_host.so")), "run __graft_entry__.build() first"
    cmd = ["g++", "-std=c++17", "-O0", "-I", os.path.join(ROOT, "visiontransformer_amd", "host", "include"),
           "-I", os.path.join(ROOT, "include"), os.path.join(REF, "app", app + ".cpp"), "-L", lib, "-lvit_host",
           "-lvitk", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o",
           str(tmp_path / app)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

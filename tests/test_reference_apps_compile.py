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


def test_vitconfig_json_hooks_work_with_the_reference_json_library(tmp_path):
    """include/model/VisionTransformer.hpp:32-57 of the reference gives ViTConfig nlohmann::json
    conversions; code that writes `json j = cfg;` / `j.get<ViTConfig>()` must keep compiling against
    the host layer's header (the reference's vendored utils/json.hpp comes from its own include
    directory, placed AFTER the host layer's), and the dump must be what save_config writes."""
    src = tmp_path / "j.cpp"
    src.write_text(r'''
#include "model/VisionTransformer.hpp"
#include "utils/ModelUtils.hpp"
#include <fstream>
#include <iostream>
#include <sstream>
int main(int, char **argv) {
  ViTConfig c;
  c.image_size = 32; c.patch_size = 8; c.in_channels = 3; c.num_classes = 7; c.embedding_dim = 48;
  c.num_heads = 6; c.num_layers = 2; c.mlp_hidden_dim = 96; c.dropout_rate = 0.25f;
  json j = c;                                  // to_json found by argument-dependent lookup
  ViTConfig d = j.get<ViTConfig>();            // from_json
  if (d.image_size != 32 || d.patch_size != 8 || d.in_channels != 3 || d.num_classes != 7 || d.embedding_dim != 48 ||
      d.num_heads != 6 || d.num_layers != 2 || d.mlp_hidden_dim != 96 || d.dropout_rate != 0.25f) return 3;
  ModelUtils::save_config(c, argv[1]);
  std::ifstream f(argv[1]);
  std::stringstream ss;
  ss << f.rdbuf();
  json k = json::parse(ss.str());
  if (k != j) return 4;
  ViTConfig e = ModelUtils::load_config(argv[1]);
  if (e.embedding_dim != 48 || e.dropout_rate != 0.25f) return 5;
  std::cout << j.dump(4) << std::endl;
  return 0;
}
''')
    lib = os.path.join(ROOT, "visiontransformer_amd", "lib")
    exe = str(tmp_path / "j")
    cmd = ["g++", "-std=c++17", "-O0", "-w", "-I", os.path.join(ROOT, "visiontransformer_amd", "host", "include"), "-I",
           os.path.join(ROOT, "include"), "-I", os.path.join(REF, "include"), str(src), "-L", lib, "-lvit_host",
           "-lvitk", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, str(tmp_path / "cfg.json")], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert '"embedding_dim": 48' in r.stdout and '"dropout_rate": 0.25' in r.stdout

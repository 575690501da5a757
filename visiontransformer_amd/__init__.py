"""visiontransformer_amd: MI355X-native (gfx950) ViT training step behind the C++ API of
PaulParizacaMozo/VisionTransformer.

The product is two native libraries built in-tree by ``make -C visiontransformer_amd``:

* ``lib/libvitk.so``     -- hand-written HIP kernels, C-ABI in ``include/vitk.h``
* ``lib/libvit_host.so`` -- C++ host layer mirroring the reference's Tensor / Layer / Loss /
  Optimizer / Trainer / ModelUtils classes, C-ABI in ``include/vit_host.h``

This Python package only binds those C-ABIs (ctypes) for the tests, ``bench.py`` and the
data-parallel driver; PyTorch supplies device memory, streams and ``torch.distributed``.
There is no CPU fallback: importing works anywhere, calling a kernel without the built
libraries or without a GPU raises.
"""
from . import host, vitk  # noqa: F401

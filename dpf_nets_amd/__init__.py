"""dpf_nets_amd -- MI355X-native hot path of dpf-nets (Discrete Point Flow Networks).

The per-point conditional affine-coupling flow decoder (reference:
lib/networks/{layers,flows,decoders,losses}.py) and the Chamfer / approximate-EMD
structural losses (reference: lib/metrics/pytorch_structural_losses) as
hand-written HIP kernels for gfx950 behind a C ABI (include/dpf_hip.h,
libdpf_hip.so), with a Python host side that mirrors the reference's module /
function surface.  See DESIGN.md and INTEGRATION.md.
"""
from ._lib import lib, lib_path, have_lib  # noqa: F401

__all__ = ["lib", "lib_path", "have_lib"]

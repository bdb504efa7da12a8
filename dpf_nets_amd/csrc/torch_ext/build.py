"""Builds StructuralLossesBackend_native*.so -- the compiled (pybind) form of the structural-losses boundary over libdpf_hip.so's
C ABI -- IN-TREE beside the Python module of the same role (dpf_nets_amd/metrics/StructuralLosses/).  Plain g++: the file has no
device code; torch's extension helper is used only for its include / library paths (CUDAExtension would run hipify over the
source, which this tree does not do).     python dpf_nets_amd/csrc/torch_ext/build.py"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
ROOT = os.path.dirname(PKG)
NAME = "StructuralLossesBackend_native"
OUT = os.path.join(PKG, "metrics", "StructuralLosses", NAME + sysconfig.get_config_var("EXT_SUFFIX"))
SRC = os.path.join(HERE, "structural_losses_backend.cpp")


def build(force=False):
    lib = os.path.join(PKG, "libdpf_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("build libdpf_hip.so first (make -C dpf_nets_amd/csrc)")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "dpf_hip.h"))):
        return OUT
    import torch
    from torch.utils import cpp_extension as CE
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + NAME,
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in CE.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]:
        cmd += ["-isystem" if "torch" in inc or rocm in inc else "-I", inc]
    cmd += [SRC, "-o", OUT, "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
            "-L" + PKG, "-ldpf_hip", "-Wl,-rpath,$ORIGIN/../..", "-Wl,-rpath," + tlib]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

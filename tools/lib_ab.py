"""(build the variants with `make -C dpf_nets_amd/csrc ablate ABLATE=<mask>`)  Run bench.py's training leg (or any bench arguments) against an alternative build of the library: lib_ab.py <so-name> [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import _lib
so = sys.argv[1]
_lib.lib_path = lambda: os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dpf_nets_amd", so)
import bench
sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()

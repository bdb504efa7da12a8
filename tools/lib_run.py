"""Run any tool of this directory against an alternative build of the library: lib_run.py <so-name> <script.py> [args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib
so = sys.argv[1]
_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", so)
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")

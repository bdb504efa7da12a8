"""pytest against an alternative build of the library: pytest_lib.py <so-name> <pytest args>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib
so = sys.argv[1]
_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", so)
import pytest
sys.exit(pytest.main(sys.argv[2:]))

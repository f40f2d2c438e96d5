"""Run another tool against a different build of the library: python tools/with_lib.py lib.so tools/x.py [args...]"""
import os, runpy, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")

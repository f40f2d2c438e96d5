"""Time one bgemm shape with another build of the library (A/B of kernel variants): python tools/bgemm_ablate.py lib.so M N K akm bkm af bf [splits]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(__file__), "bgemm_one.py"), run_name="__main__")

"""A/B a kernel micro-benchmark between library builds: python tools/ab_lib.py <lib.so> <bench names...>"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_kernels.py"), run_name="__main__")

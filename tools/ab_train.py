"""Run tools/bench_train.py against another library build: python tools/ab_train.py <lib.so> [bench_train args...]"""
import os, sys, runpy
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from css_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

"""bench.py against another build of the library: MAAVSS_LIB=<path to .so> python scripts/ab_step.py [bench.py flags]"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from maavss_amd import _lib
if os.environ.get("MAAVSS_LIB"):
    _lib.LIB_PATH = os.environ["MAAVSS_LIB"]
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

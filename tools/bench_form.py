"""bench.py with a forced block-sparse kernel form: python tools/bench_form.py FORM [bench.py arguments]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sed-net_amd")]
from sednet_hip import ops
ops.MS_SPARSE_FORM = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

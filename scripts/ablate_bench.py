"""bench.py under a test knob of the convolution library (mink_conv_set_stagger bits; 192 = the mid-layer weight gradients
without their gathers and matrix work): what a step would take if a family of kernels were free.
usage: python scripts/ablate_bench.py <knob> [bench.py arguments]   (results are WRONG by construction: timing only)"""
import sys, runpy
sys.path.insert(0, ".")
from nerf_downstream_amd._lib import lib
lib().mink_conv_set_stagger(int(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")

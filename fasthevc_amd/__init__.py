"""fasthevc_amd -- MI355X-native CU-partition fast-decision path (HM TEncCu depth-map predictor).

The compute path is the HIP library built from fasthevc_amd/csrc (C ABI: include/fasthevc.h); Python is
only the host-side mirror used by tests, the trainer and bench.py.  There is no CPU fallback.
"""
__version__ = "0.1.0"

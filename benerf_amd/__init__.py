"""benerf_amd - MI355X-native (gfx950) hot path of BeNeRF behind the reference's own
Python API.  csrc/ holds the hand-written HIP kernels + C ABI (include/benerf_hip.h);
the modules next to this file mirror the reference's operator interface
(spline, run_nerf_helpers, model.nerf, model.optimize, ...) and dispatch to the kernels.
"""
__version__ = "0.1.0"

"""benerf_amd - MI355X-native (gfx950) hot path of BeNeRF behind the reference's own
Python API.  csrc/ holds the hand-written HIP kernels + C ABI (include/benerf_hip.h);
the modules next to this file mirror the reference's operator interface
(spline, run_nerf_helpers, model.nerf, model.optimize, ...) and dispatch to the kernels.
"""
import os as _os

# The training step uses up to five HIP streams at once (main, the dW side stream, the caller's input loader, RCCL's
# communicator streams).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): two streams
# that land on one queue run IN ORDER - with RCCL streams present the main and the dW stream did, and a data-parallel step lost
# 0.2 ms to it (profiles/r03_rccl_loopback.log).  Read when the runtime initialises, i.e. at the first device call: set here
# unless the user chose a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"

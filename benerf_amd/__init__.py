"""benerf_amd - MI355X-native (gfx950) hot path of BeNeRF behind the reference's own
Python API.  csrc/ holds the hand-written HIP kernels + C ABI (include/benerf_hip.h);
the modules next to this file mirror the reference's operator interface
(spline, run_nerf_helpers, model.nerf, model.optimize, ...) and dispatch to the kernels.
"""
import os as _os
import warnings as _warnings

HW_QUEUES_WANTED = 8


def configure_runtime(hw_queues=HW_QUEUES_WANTED):
    """Opt-in process-wide HIP runtime setting for multi-stream training: GPU_MAX_HW_QUEUES (default 4 in the runtime).

    A training step keeps up to five HIP streams busy at once (main, the dW side stream, the caller's input loader, RCCL's
    communicator streams).  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues; two streams that land on one
    queue run IN ORDER - with RCCL's streams present the main and the dW stream did, and a data-parallel step lost 0.2 ms to it
    (profiles/r03_rccl_loopback.log).  The variable is read when the runtime initialises (the first device call), so call this
    BEFORE touching the GPU; it never overrides a value the user has set.  Importing benerf_amd does NOT call it (a library must
    not change its host process' runtime behind the caller's back) unless BENERF_SET_HW_QUEUES=1 is in the environment;
    `bench.py` calls it, `engine.TrainStep` warns once when it is handed a process group and finds fewer than 6 queues.
    Returns the value in effect (as far as the environment tells)."""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    return hw_queues_in_effect()


def hw_queues_in_effect():
    """GPU_MAX_HW_QUEUES as the environment has it (the HIP runtime's default is 4 when it is unset)."""
    try:
        return int(_os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


_warned_queues = False


def warn_if_few_hw_queues(minimum=6):
    global _warned_queues
    if not _warned_queues and hw_queues_in_effect() < minimum:
        _warned_queues = True
        _warnings.warn("benerf_amd: data-parallel TrainStep with GPU_MAX_HW_QUEUES=%d: the step's streams (main, dW, RCCL) may share a "
                       "hardware queue and run in order (+0.2 ms per step measured); call benerf_amd.configure_runtime() before the "
                       "first GPU call, or export GPU_MAX_HW_QUEUES=8" % hw_queues_in_effect(), RuntimeWarning, stacklevel=3)


if _os.environ.get("BENERF_SET_HW_QUEUES") == "1":
    configure_runtime()

__version__ = "0.1.0"

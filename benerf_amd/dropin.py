"""Makes reference-style drivers (`from model.nerf import *`, `import spline`,
`from run_nerf_helpers import ...`) resolve to this package: registers the mirrored modules in
sys.modules under the reference's top-level names.  See INTEGRATION.md."""
import importlib
import sys

_ALIASES = {
    "spline": "benerf_amd.spline",
    "bezier": "benerf_amd.bezier",
    "run_nerf_helpers": "benerf_amd.run_nerf_helpers",
    "model": "benerf_amd.model",
    "model.nerf": "benerf_amd.model.nerf",
    "model.optimize": "benerf_amd.model.optimize",
    "model.embedder": "benerf_amd.model.embedder",
    "model.component": "benerf_amd.model.component",
    "loss": "benerf_amd.loss",
    "loss.imgloss": "benerf_amd.loss.imgloss",
    "utils": "benerf_amd.utils",
    "utils.math_utils": "benerf_amd.utils.math_utils",
    "utils.img_utils": "benerf_amd.utils.img_utils",
    "utils.event_utils": "benerf_amd.utils.event_utils",
}


def install():
    """After this call `import spline`, `from model import optimize`, ... import benerf_amd's
    HIP-backed modules.  Returns the list of installed names."""
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
    return sorted(_ALIASES)

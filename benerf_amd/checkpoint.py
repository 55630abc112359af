"""Checkpoint read / write in the reference's format (train.py:443-455 writes, test.py:98-107 reads): one
`torch.save` dict with `global_step`, the `graph` state-dict and the five Adam state-dicts, so checkpoints move
between the reference and this path in both directions.  The fused `engine.TrainStep` keeps its Adam moments in
flat buffers; `TrainStep.export_optimizer_state / import_optimizer_state` translate to and from the
`torch.optim.Adam` objects `Model.setup_optimizer` returns."""
import torch

OPTIMIZER_KEYS = ("optimizer_nerf", "optimizer_pose", "optimizer_trans", "optimizer_rgb_crf", "optimizer_event_crf")


def save(path, graph, optimizers, global_step):
    """optimizers: the 5-tuple of Model.setup_optimizer (nerf, pose, transform, rgb_crf, event_crf)."""
    assert len(optimizers) == len(OPTIMIZER_KEYS)
    ck = {"global_step": int(global_step), "graph": graph.state_dict()}
    for key, opt in zip(OPTIMIZER_KEYS, optimizers):
        ck[key] = opt.state_dict()
    torch.save(ck, path)


def load(path, graph, optimizers=None, map_location=None):
    """Restores `graph` (and the optimisers when given) in place; returns global_step.  Parameters keep their
    storage (load_state_dict copies into it), so a TrainStep built on the graph sees the loaded values."""
    ck = torch.load(path, map_location=map_location)
    graph.load_state_dict(ck["graph"])
    if optimizers is not None:
        for key, opt in zip(OPTIMIZER_KEYS, optimizers):
            opt.load_state_dict(ck[key])
    return int(ck["global_step"])

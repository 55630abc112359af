"""Host-side mirror of utils/event_utils.py: the one function on the path."""
import numpy as np
import torch

from .. import kernels as K


@torch.no_grad()
def accumulate_events_on_gpu(out, xs, ys, ps):
    """out[y, x] += p with duplicates summed (utils/event_utils.py:246-259) as float atomics in
    kernel K7 (polarities are +-1 => exact, order independent).  Returns a float64 device tensor
    like the reference (np.zeros float64 + float32 dense)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    H, W = out.shape[0], out.shape[1]
    acc = torch.as_tensor(np.asarray(out), dtype=torch.float32, device=dev).contiguous().clone()
    x = torch.as_tensor(np.asarray(xs).astype(np.int32), device=dev)
    y = torch.as_tensor(np.asarray(ys).astype(np.int32), device=dev)
    p = torch.as_tensor(np.asarray(ps).astype(np.float32), device=dev)
    K.event_accumulate(x, y, p, H, W, out=acc)
    return acc.double()

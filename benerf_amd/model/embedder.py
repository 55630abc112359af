"""Host-side mirror of the reference's model/embedder.py.  Inside the network the encoding is
fused into the MLP kernel (never written to HBM); this stand-alone form backs
`get_embedder(...)` for callers that want the encoding itself."""
import torch
from torch import nn

from .. import kernels as K


class Embedder:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]  (model/embedder.py:4-34)."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        if not kwargs.get("log_sampling", True):
            raise NotImplementedError("only log-sampled frequency bands (the reference's setting) are implemented")
        fns = kwargs.get("periodic_fns", [torch.sin, torch.cos])
        if list(fns) != [torch.sin, torch.cos]:
            raise NotImplementedError("periodic_fns must be [sin, cos]")
        self.n_freqs = int(kwargs["num_freqs"])
        if int(kwargs["max_freq_log2"]) != self.n_freqs - 1:
            raise NotImplementedError("max_freq_log2 must equal num_freqs - 1")
        self.include_input = bool(kwargs["include_input"])
        d = int(kwargs["input_dims"])
        self.out_dim = (d if self.include_input else 0) + 2 * d * self.n_freqs

    def embed(self, inputs):
        return K.posenc(inputs.detach().float().contiguous(), self.n_freqs, self.include_input)


def get_embedder(args, multires, i=0):
    """(embed_fn, out_dim)  (model/embedder.py:37-52)."""
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=not args.use_barf_c2f, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim

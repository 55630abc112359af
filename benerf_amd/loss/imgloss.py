"""Host-side mirror of loss/imgloss.py: MSELoss as a HIP reduction with autograd.  (The fused
training step computes all loss terms and their gradients in kernel K6 instead.)"""
import torch

from .. import kernels as K


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ac, bc = a.detach().float().contiguous(), b.detach().float().contiguous()
        ctx.save_for_backward(ac, bc)
        ctx.shapes = (a.shape, b.shape, a.dtype, b.dtype)
        return K.mse_fwd(ac.view(-1), bc.view(-1))

    @staticmethod
    def backward(ctx, g):
        ac, bc = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad
        da, db = K.mse_bwd(ac.view(-1), bc.view(-1), g.reshape(1).float().contiguous(), need_a, need_b)
        sa, sb, ta, tb = ctx.shapes
        return (None if da is None else da.view(sa).to(ta)), (None if db is None else db.view(sb).to(tb))


class MSELoss:
    """mean((img - gt)**2)  (loss/imgloss.py:3-5)."""

    def __call__(self, img, gt_img):
        if img.shape != gt_img.shape:
            img, gt_img = torch.broadcast_tensors(img, gt_img)
        return _Mse.apply(img, gt_img)

"""`torch.optim.Adam` objects whose `step()` is ONE fused K8 launch (benerf_adam_step) over a flat parameter arena.

The reference builds five `torch.optim.Adam` (model/optimize.py:36-55) and its training loop calls `zero_grad()`, `step()` and
writes `param_group["lr"]` on them (train.py:196-200, 343-394); checkpoints store their `state_dict()` (train.py:443-455).
`FlatAdam` IS a `torch.optim.Adam` - same constructor, `param_groups`, `state`, `state_dict()` / `load_state_dict()` format
(per parameter: `step`, `exp_avg`, `exp_avg_sq`) - but

  * at its first `step()` its parameters are re-homed into one flat fp32 buffer (`ParamArena`: the `nn.Parameter`s keep names
    and shapes and alias it), next to flat gradient / exp_avg / exp_avg_sq buffers of the same layout;
  * the render nodes (engine.RenderPair / RenderRays) write the weight gradients straight into the arena's gradient buffer and
    hand autograd VIEWS of it, which `AccumulateGrad` adopts as `.grad` without a copy (the gradient is `None` after the
    reference's `zero_grad()`), so `p.grad` is a window on the arena;
  * `step()` checks those aliases (pointer comparisons, no device work), repairs what a caller replaced (a `.grad` that is
    somebody else's tensor is copied in; state loaded by `load_state_dict` is adopted) and updates every parameter of a group
    with one launch - torch's own multi-tensor Adam costs ~10 launches per optimiser and, under the reference's
    `torch.set_default_tensor_type("torch.cuda.FloatTensor")` (train.py:472), one `.item()` synchronisation PER PARAMETER and
    step, because its `step` counters are then created on the device (96 per iteration for the two NeRFs).

Arithmetic = torch's Adam with default betas / eps (benerf_adam_step is pinned by golden vector G10).  Options that change the
update rule (weight_decay, amsgrad, maximize) are refused: the reference never sets them.
"""
import torch

from . import kernels as K


class ParamArena:
    """Flat fp32 storage for a list of parameters: value, gradient, exp_avg, exp_avg_sq, all with one layout."""

    def __init__(self, params):
        params = list(params)
        dev = params[0].device
        self.device = dev
        self.params = params
        n = sum(p.numel() for p in params)
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.slots = []                     # (offset, numel) per parameter, in order
        self.index = {id(p): i for i, p in enumerate(params)}
        o = 0
        for p in params:
            if p.dtype != torch.float32:
                raise TypeError("ParamArena: float32 parameters only")
            k = p.numel()
            self.slots.append((o, k))
            self.adopt(p, o, k)
            o += k
        self.base_p, self.base_g = self.p.data_ptr(), self.g.data_ptr()
        # which backward pass (torch._C._current_graph_task_id) last wrote the gradient slots starting at parameter i0: a second
        # render node of the SAME pass must add on top of the first one's result instead of overwriting it (engine._grad_targets)
        self.pass_ids = {}

    def adopt(self, p, o, k):
        """Move p's value into its slot and make p alias it (names, shapes, autograd identity unchanged)."""
        view = self.p[o:o + k].view(p.shape)
        view.copy_(p.detach())
        p.data = view
        p._benerf_slot = (self, o, k)

    def homed(self, i):
        p = self.params[i]
        o, k = self.slots[i]
        return p.data_ptr() == self.base_p + 4 * o and p.is_contiguous()

    def grad_view(self, i):
        """A FRESH view of parameter i's gradient slot (nobody else holds it: autograd may adopt it as `.grad`)."""
        o, k = self.slots[i]
        return self.g[o:o + k].view(self.params[i].shape)

    def grad_is_window(self, i):
        """True if parameter i's `.grad` is a window on its slot of the gradient buffer."""
        g = self.params[i].grad
        o, k = self.slots[i]
        return g is not None and g.data_ptr() == self.base_g + 4 * o and g.is_contiguous() and g.dtype == torch.float32


def arena_of(params):
    """(arena, [slot index of each parameter]) if all of `params` are homed members of ONE arena; else None."""
    first = getattr(params[0], "_benerf_slot", None)
    if first is None:
        return None
    arena = first[0]
    idx = []
    for p in params:
        slot = getattr(p, "_benerf_slot", None)
        if slot is None or slot[0] is not arena:
            return None
        i = arena.index.get(id(p))
        if i is None or not arena.homed(i):
            return None
        idx.append(i)
    return arena, idx


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, **kw):
        super().__init__(params, lr=lr, **kw)
        # The arena is built - and the parameters re-homed into it - at the FIRST step(), not here: an optimiser that never steps
        # (engine.TrainStep keeps its own flat buffers and uses these objects only as checkpoint containers, export / import
        # _optimizer_state) must not move anybody's storage.  The first iteration of a loop therefore runs the generic path
        # (fresh gradient tensors, copied in), every later one the in-place path.
        self._arena = None
        self._index = {}
        self._state_seen = {}               # id(p) -> id(state dict) whose tensors are known to alias the arena

    # -- state: the reference's checkpoint format, backed by the arena -------------------------------------------------------
    def _adopt_state(self, p, i):
        """state[p] as torch's Adam keeps it, with exp_avg / exp_avg_sq as views of the arena; values set by load_state_dict or
        by anybody who assigned `state[p]` are copied in first."""
        a = self._arena
        o, k = a.slots[i]
        st = self.state.get(p)
        m_view, v_view = a.m[o:o + k].view(p.shape), a.v[o:o + k].view(p.shape)
        if not st:
            m_view.zero_()
            v_view.zero_()
            st = {"step": torch.tensor(0.0, dtype=torch.float32, device="cpu"), "exp_avg": m_view, "exp_avg_sq": v_view}
            self.state[p] = st
        else:
            for key, view in (("exp_avg", m_view), ("exp_avg_sq", v_view)):
                cur = st.get(key)
                if cur is None:
                    view.zero_()
                elif cur.data_ptr() != view.data_ptr():
                    view.copy_(cur.to(device=view.device, dtype=torch.float32).reshape(view.shape))
                st[key] = view
            step = st.get("step", 0.0)
            step = float(step.item()) if torch.is_tensor(step) else float(step)     # a loaded device counter: one read, once
            st["step"] = torch.tensor(step, dtype=torch.float32, device="cpu")
        self._state_seen[id(p)] = id(st)
        return st

    def _live_state(self, p, i):
        st = self.state.get(p)
        if st and self._state_seen.get(id(p)) == id(st) and st["exp_avg"].data_ptr() == self._arena.m.data_ptr() + 4 * self._arena.slots[i][0]:
            return st
        return self._adopt_state(p, i)

    def state_dict(self):
        # every parameter that has been stepped carries live state already; nothing to flush
        return super().state_dict()

    def zero_grad(self, set_to_none=True):
        if self._arena is not None:
            self._arena.pass_ids.clear()
        return super().zero_grad(set_to_none=set_to_none)

    # -- the update ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a = self._arena
        if a is None:
            flat = [p for g in self.param_groups for p in g["params"]]
            if not flat:
                return loss
            a = self._arena = ParamArena(flat)
            self._index = a.index
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("benerf_amd.optim.FlatAdam implements the reference's Adam (model/optimize.py:36-55): no "
                                          "weight_decay / amsgrad / maximize")
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            run = None                      # [first offset, numel, t] of a run of consecutive slots updated by one launch
            steps = []
            for p in group["params"]:
                if p.grad is None:          # torch's Adam leaves such a parameter (and its counters) alone
                    run = self._flush(run, lr, beta1, beta2, eps)
                    continue
                i = self._index.get(id(p))
                if i is None:
                    raise NotImplementedError("FlatAdam: a parameter added after the first step() (add_param_group) has no slot in the arena")
                o, k = a.slots[i]
                if not a.grad_is_window(i):
                    a.g[o:o + k].copy_(p.grad.detach().to(torch.float32).reshape(-1))
                st = self._live_state(p, i)
                t = int(st["step"].item()) + 1          # a host tensor: the checkpointed counter is the only counter
                steps.append(st["step"])
                if not a.homed(i):
                    # somebody re-pointed p.data (e.g. engine.TrainStep took the graph over and owns the storage now): update the
                    # parameter where it lives, moments in this arena - never steal the storage back behind the other owner
                    run = self._flush(run, lr, beta1, beta2, eps)
                    if not p.data.is_contiguous():
                        raise RuntimeError("FlatAdam: a parameter outside the arena must be contiguous")
                    K.adam_step(p.data.view(-1), a.g[o:o + k], a.m[o:o + k], a.v[o:o + k], lr, t, beta1, beta2, eps)
                    continue
                if run is not None and run[0] + run[1] == o and run[2] == t:
                    run[1] += k
                else:
                    self._flush(run, lr, beta1, beta2, eps)
                    run = [o, k, t]
            self._flush(run, lr, beta1, beta2, eps)
            if steps:
                torch._foreach_add_(steps, 1.0)
        a.pass_ids.clear()
        return loss

    def _flush(self, run, lr, beta1, beta2, eps):
        if run is not None:
            a = self._arena
            o, k, t = run
            K.adam_step(a.p[o:o + k], a.g[o:o + k], a.m[o:o + k], a.v[o:o + k], lr, t, beta1, beta2, eps)
        return None

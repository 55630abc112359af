"""Host-side mirror of the reference's model/optimize.py: `Model` (network + optimiser
construction) and the trajectory queries `Graph.get_pose_evt / get_pose_rgb`, which dispatch
to the SE(3) spline kernel K1 (autograd-enabled) instead of ~1300 tiny ATen launches."""
import torch

from .. import engine
from . import nerf
from .component import ColorToneMapper, LuminanceToneMapper
from .component import ControlKnotLieAlgebra, TransformationLieAlgebra
from ..optim import FlatAdam


class Model(nerf.Model):
    def __init__(self, args):
        # the reference hard-codes this network shape (model/optimize.py:8-9)
        self.graph = Graph(args, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)

    def build_network(self, args, poses=None, event_poses=None):
        """Adds the trajectory / transform / CRF members and initialises them
        (model/optimize.py:11-34): knots ~ U(0, 0.01), transform = 0."""
        g = self.graph
        g.evt_knot_pose_se3 = ControlKnotLieAlgebra(4)
        g.rgb_knot_pose_se3 = ControlKnotLieAlgebra(4)     # unused by the path, kept for checkpoints
        g.transform = TransformationLieAlgebra(1)
        g.rgb_crf = ColorToneMapper(hidden=args.rgb_crf_net_hidden, width=args.rgb_crf_net_width, input_type="Gray")
        g.event_crf = LuminanceToneMapper(hidden=args.event_crf_net_hidden, width=args.event_crf_net_width,
                                          input_type="Gray")
        dev = g.nerf.alpha_linear.weight.device
        knots = torch.cat([torch.rand(1, 6) * 0.01 for _ in range(4)]).to(dev)
        g.evt_knot_pose_se3.params.weight.data = knots
        g.transform.params.weight.data = torch.zeros(1, 6, device=dev)
        g.rgb_crf.weights_biases_init()
        g.event_crf.weights_biases_init()
        g.to(dev)
        return g

    def setup_optimizer(self, args):
        """Five Adam optimisers with torch defaults (model/optimize.py:36-55).  They ARE torch.optim.Adam objects (param_groups,
        state, state_dict format, zero_grad) whose step() is one fused launch over a flat parameter arena: benerf_amd/optim.py."""
        g = self.graph
        # the reference's parameter order: a checkpoint's optimizer state_dict maps state to parameters BY POSITION (train.py:443-455)
        grad_vars = list(g.nerf.parameters())
        if args.N_importance > 0:
            grad_vars += list(g.nerf_fine.parameters())
        self.optim_nerf = FlatAdam(params=grad_vars, lr=args.lrate)
        self.optim_pose = FlatAdam(params=list(g.evt_knot_pose_se3.parameters()), lr=args.pose_lrate)
        self.optim_transform = FlatAdam(params=list(g.transform.parameters()), lr=args.transform_lrate)
        self.optim_event_crf = FlatAdam(params=list(g.event_crf.mlp_luminance.parameters()), lr=args.event_crf_lrate)
        self.optim_rgb_crf = FlatAdam(params=list(g.rgb_crf.mlp_gray.parameters()), lr=args.rgb_crf_lrate)
        return self.optim_nerf, self.optim_pose, self.optim_transform, self.optim_rgb_crf, self.optim_event_crf


class Graph(nerf.Graph):
    def _poses(self, args, ts, n_poses, with_transform):
        knots = self.evt_knot_pose_se3.params.weight
        dev = knots.device
        if not torch.is_tensor(ts):
            ts = torch.tensor([float(ts[0]), float(ts[1])], dtype=torch.float32)
        ts2 = ts.reshape(-1)[:2].to(device=dev, dtype=torch.float32)
        traj = {"spline": 0, "linear": 1}[args.traj]
        transform = self.transform.params.weight if with_transform else None
        return engine.SplinePoses.apply(knots, transform, ts2, int(n_poses), traj, False)

    def get_pose_evt(self, args, events_ts, seg_num=None):
        """Event-camera poses at linspace(ts[0], ts[1], seg_num or 2)  (model/optimize.py:58-82)."""
        return self._poses(args, events_ts, 2 if seg_num is None else seg_num, False)

    def get_pose_rgb(self, args, exposure_ts, seg_num=None):
        """RGB-camera poses: knots + transform in se(3), linspace over the exposure
        (model/optimize.py:84-111)."""
        return self._poses(args, exposure_ts, args.num_interpolated_pose if seg_num is None else seg_num, True)

    _stock_pose_queries = {"get_pose_evt": get_pose_evt, "get_pose_rgb": get_pose_rgb}

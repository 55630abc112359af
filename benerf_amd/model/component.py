"""Parameter containers with the reference's state-dict names (model/component.py:7-15) and
the optional tone-mapper heads (model/component.py:38-149; `optimize_*_crf = False` in every
shipped config, so they stay small torch modules outside the HIP hot path)."""
import torch
import torch.nn as nn
import torch.nn.init as init


class ControlKnotLieAlgebra(nn.Module):
    """4 se(3) control knots: `params.weight` [knot_num, 6]  (model/component.py:7-10)."""

    def __init__(self, knot_num):
        super().__init__()
        self.params = nn.Embedding(knot_num, 6)


class TransformationLieAlgebra(nn.Module):
    """event->rgb camera offset in se(3): `params.weight` [trans_num, 6]  (model/component.py:12-15)."""

    def __init__(self, trans_num):
        super().__init__()
        self.params = nn.Embedding(trans_num, 6)


class ExposureTime(nn.Module):
    def __init__(self):
        super().__init__()
        self.params = nn.Embedding(2, 1)


def _tone_mlp(in_dim, width, hidden):
    layers = [nn.Linear(in_dim, width), nn.ReLU()]
    for _ in range(hidden):
        layers += [nn.Linear(width, width), nn.ReLU()]
    layers.append(nn.Linear(width, 1))
    return nn.Sequential(*layers)


class ColorToneMapper(nn.Module):
    """1 -> width -> 1 MLP + sigmoid per colour; keys `mlp_gray.{0,2}.*`  (model/component.py:38-110)."""

    def __init__(self, hidden=0, width=128, input_type="Gray"):
        super().__init__()
        self.net_hidden, self.net_width, self.input_type = hidden, width, str(input_type)
        if self.input_type == "Gray":
            self.mlp_gray = _tone_mlp(1, width, hidden)
        else:
            shared = _tone_mlp(1, width, hidden)      # the reference reuses ONE layer list for r, g, b
            self.mlp_r = shared
            self.mlp_g = shared
            self.mlp_b = shared

    def _nets(self):
        return [self.mlp_gray] if self.input_type == "Gray" else [self.mlp_r, self.mlp_g, self.mlp_b]

    def weights_biases_init(self):
        for net in self._nets():
            for layer in net:
                if isinstance(layer, nn.Linear):
                    init.xavier_uniform_(layer.weight)
                    init.zeros_(layer.bias)

    def forward(self, radience):
        if self.input_type == "Gray":
            raw = self.mlp_gray(radience)
        else:
            raw = torch.cat([self.mlp_r(radience[:, 0]), self.mlp_g(radience[:, 1]), self.mlp_b(radience[:, 2])], -1)
        return torch.sigmoid(raw)

    def constraint_radience_scale(self, fixed_value=0.5):
        zero = torch.zeros(1, device=next(self.parameters()).device)
        return torch.mean((torch.sigmoid(self.mlp_gray(zero)) - fixed_value) ** 2)


class LuminanceToneMapper(nn.Module):
    """keys `mlp_luminance.{0,2}.*`; biases initialised to ONE  (model/component.py:112-149)."""

    def __init__(self, hidden=0, width=128, input_type="Gray"):
        super().__init__()
        self.net_hidden, self.net_width, self.input_type = hidden, width, str(input_type)
        self.mlp_luminance = _tone_mlp(1 if self.input_type == "Gray" else 3, width, hidden)

    def weights_biases_init(self):
        for layer in self.mlp_luminance:
            if isinstance(layer, nn.Linear):
                init.xavier_uniform_(layer.weight)
                init.ones_(layer.bias)

    def forward(self, radience):
        return torch.sigmoid(self.mlp_luminance(radience))

"""Activation factory (reference: `basic/activation.py:27-54`).

relu / sigmoid / softmax run fused inside the HIP affine+activation kernel; the returned nn.Module
only marks the slot in `MLP.mlp` (it holds no parameters, so the state_dict layout is unchanged).
Dice / PReLU / LeakyReLU are outside the hot path: they stay plain torch modules
applied after the fused BatchNorm affine."""
import torch
import torch.nn as nn

FUSED = ("relu", "sigmoid", "softmax")


class Dice(nn.Module):
    """DIN's Dice (`basic/activation.py:5-25`), kept for API completeness; not on the hot path."""

    def __init__(self, epsilon=1e-3):
        super().__init__()
        self.epsilon = epsilon
        self.alpha = nn.Parameter(torch.randn(1))

    def forward(self, x):
        avg = x.mean(dim=1, keepdim=True)
        var = (torch.pow(x - avg, 2) + self.epsilon).sum(dim=1, keepdim=True)
        ps = torch.sigmoid((x - avg) / torch.sqrt(var))
        return ps * x + (1 - ps) * self.alpha * x


def activation_name(act):
    """Lower-case name when `act` is one of the fused activations, else None."""
    return act.lower() if isinstance(act, str) and act.lower() in FUSED else None


def activation_layer(act_name):
    if isinstance(act_name, str):
        table = {"sigmoid": nn.Sigmoid, "relu": lambda: nn.ReLU(inplace=True), "dice": Dice, "prelu": nn.PReLU,
                 "softmax": lambda: nn.Softmax(dim=1), "leakyrelu": lambda: nn.LeakyReLU(0.1)}
        if act_name.lower() not in table:
            raise NotImplementedError(act_name)
        return table[act_name.lower()]()
    if isinstance(act_name, type) and issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError

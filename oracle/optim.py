"""torch.optim.Adam restated (oracle only).

The reference trains with `torch.optim.Adam(model.parameters(), lr,
weight_decay)` (`trainers/ctr_trainer.py:50-52,73`): betas (0.9, 0.999),
eps 1e-8, L2 weight decay ADDED TO THE GRADIENT (not AdamW), no amsgrad.
Parameters whose gradient is None are skipped entirely (no decay, no state).
"""
import numpy as np


class Adam:
    def __init__(self, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.wd, self.b1, self.b2, self.eps = lr, weight_decay, betas[0], betas[1], eps
        self.state = {}

    def step(self, params, grads):
        """In-place update of `params[name]` for every name in `grads`."""
        for name, g in grads.items():
            p = params[name]
            st = self.state.setdefault(name, {"step": 0, "m": np.zeros_like(p), "v": np.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            g = g.astype(p.dtype)
            if self.wd != 0:
                g = g + self.wd * p
            st["m"] = st["m"] + (g - st["m"]) * p.dtype.type(1 - self.b1)       # lerp_
            st["v"] = st["v"] * p.dtype.type(self.b2) + p.dtype.type(1 - self.b2) * g * g
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            step_size = self.lr / bc1
            denom = np.sqrt(st["v"]) / p.dtype.type(np.sqrt(bc2)) + p.dtype.type(self.eps)
            params[name] = (p - p.dtype.type(step_size) * (st["m"] / denom)).astype(p.dtype)

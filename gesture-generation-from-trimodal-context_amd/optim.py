"""Fused Adam over a flat parameter slab (torch.optim.Adam semantics, train.py:104-109)."""
import torch

from . import ops


class FusedAdam:
    """One launch per step for the whole network.  Same maths as torch.optim.Adam(lr, betas, eps=1e-8) without weight
    decay / amsgrad; the step counter lives on the device so a captured hipGraph replays correctly."""

    def __init__(self, module_or_engine, lr, betas=(0.5, 0.999), eps=1e-8):
        eng = getattr(module_or_engine, "engine", module_or_engine)
        self.engine = eng
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)

    @property
    def slab(self):
        return self.engine.slab.ensure()

    def zero_grad(self, set_to_none=False):
        self.slab.zero_grad()

    def step(self, counter_advanced=False):
        """counter_advanced: the device-side step counter was already incremented for this step (ops.iter_begin at the iteration's start)."""
        s = self.slab
        n = s.n_train                                  # frozen parameters sit behind the trainable prefix and are never stepped
        if not counter_advanced:
            ops.counter_inc(s.step)
        ops.adam_step(s.flat[:n], s.grad[:n], s.m[:n], s.v[:n], self.lr, self.betas[0], self.betas[1], self.eps, s.step)

    def state_dict(self):
        s = self.slab
        return {"step": int(s.step.item()), "exp_avg": s.m.clone(), "exp_avg_sq": s.v.clone(), "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "names": list(s.names), "offsets": list(s.offsets)}

    def load_state_dict(self, sd):
        s = self.slab
        # the moments are flat images of the slab: they only mean something under the layout they were saved with (the order depends on
        # which parameters were frozen, e.g. freeze_wordembed)
        if "names" in sd and (list(sd["names"]) != list(s.names) or list(sd["offsets"]) != list(s.offsets)):
            raise ValueError("optimizer state was saved under a different parameter layout (different frozen set or network): "
                             f"{len(sd['names'])} tensors / {len(s.names)} here")
        assert sd["exp_avg"].numel() == s.m.numel(), (sd["exp_avg"].numel(), s.m.numel())
        s.m.copy_(sd["exp_avg"]); s.v.copy_(sd["exp_avg_sq"])
        s.step.fill_(int(sd["step"]))

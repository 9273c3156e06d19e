"""Fused Adam over the model's flat parameter buffer (one HIP launch per step).

State-dict compatible with `torch.optim.Adam` (reference trainer.py:109-122,829-837): per-parameter
`step`, `exp_avg`, `exp_avg_sq` entries are views of two flat moment buffers."""
from __future__ import annotations

import torch

from . import _lib, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._model = model
        self._steps = 0
        self._m = self._v = None

    def _moments(self):
        flat, _ = self._model.flat_parameters()
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != flat.device:
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        return self._m, self._v

    @torch.no_grad()
    def step(self, closure=None):
        flat, grad = self._model.flat_parameters()
        if not flat.is_cuda:
            raise _lib.RacError("FusedAdam runs on the GPU only")
        m, v = self._moments()
        g = self.param_groups[0]
        self._steps += 1
        # one pass that also writes the next step's operand parts of the split-precision conv weights, when their parts
        # and maxima are current (every step after the first); plain Adam over the whole buffer otherwise
        if ops.fused_adam_step(flat, grad, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._steps):
            return
        ops.PARAM_EPOCH += 1  # invalidates caches derived from the parameters (padded weight copies, operand parts)
        _lib.call("rac_adam_step", flat.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), flat.numel(),
                  float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self._steps,
                  _lib.stream_ptr())

    def _views(self):
        m, v = self._moments()
        for p in self.param_groups[0]["params"]:
            off = p._rac_off  # the parameter's place in the model's flat buffer (SVGConvModel._flatten)
            yield p, torch.as_strided(m, p.shape, p.stride(), off), torch.as_strided(v, p.shape, p.stride(), off)

    def state_dict(self):
        if self._steps:
            for p, mv, vv in self._views():
                self.state[p] = {"step": torch.tensor(float(self._steps)), "exp_avg": mv, "exp_avg_sq": vv}
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = 0
        for p, mv, vv in self._views():
            st = self.state.get(p)
            if st:
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
                steps = max(steps, int(st["step"]))
        self._steps = steps
        self.state.clear()

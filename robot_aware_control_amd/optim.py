"""Fused Adam over the model's flat parameter buffer (one HIP launch per step).

State-dict compatible with `torch.optim.Adam` (reference trainer.py:109-122,829-837): per-parameter
`step`, `exp_avg`, `exp_avg_sq` entries are views of two flat moment buffers."""
from __future__ import annotations

import os

import torch

import torch.distributed as dist

from . import _lib, ops


class FusedAdam(torch.optim.Optimizer):
    # The pass is bound by HBM (8.6 GB per step at g512: ~1.7 ms); the forward pass that follows starts with the encoder's
    # many small launches (2.2 ms, bound by latency, reading only the encoder's parameters).  With `overlap_next_forward` the
    # update of the large conv weights BEHIND the encoder's (the ConvLSTMs', the decoder's: 90 % of the bytes) runs on a side
    # stream and the model waits for it behind its encoder (SVGConvModel._encode through ops.PARAM_GATE, as for the sharded
    # optimiser's all-gather): same arithmetic, same bits.  Until that wait the late weights hold old values on the main
    # stream -- so this is OFF unless a training loop that owns every reader of the parameters turns it on
    # (PredictionTrainer.train, bench.py); `state_dict`, `load_state_dict` and `wait_params()` wait.
    overlap_next_forward = False

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._model = model
        self._steps = 0
        self._m = self._v = None
        self._late = []  # [(event, [(offset, numel)])] of the late weights' updates in flight, in the order they complete

    def wait_params(self, upto: int = None):
        """Make the current stream wait for the late weights' update (`upto`: only if a late weight lies below that flat
        element -- none does: the late groups start behind the encoder)."""
        if not self._late:
            return
        if upto is not None and all(off >= upto for _, spans in self._late for off, _ in spans):
            return
        torch.cuda.current_stream().wait_event(self._late[-1][0])  # (one in-order side stream: the last event covers all)
        self._late = []
        if ops.PARAM_GATE is self:
            ops.PARAM_GATE = None

    def wait_for(self, t: torch.Tensor):
        """... for the late group that holds the parameter memory `t` (and the groups in front of it: one stream)."""
        if not self._late:
            return
        flat, _ = self._model.flat_parameters()
        lo = (t.data_ptr() - flat.data_ptr()) // 4
        if lo < 0 or lo >= flat.numel():
            return  # not a view of this model's flat parameter buffer
        hi = lo + t.numel()
        hit = [k for k, (_, spans) in enumerate(self._late) if any(off < hi and lo < off + n for off, n in spans)]
        if not hit:
            return
        torch.cuda.current_stream().wait_event(self._late[hit[-1]][0])
        self._late = self._late[hit[-1] + 1:]
        if not self._late and ops.PARAM_GATE is self:
            ops.PARAM_GATE = None

    def ready(self, t: torch.Tensor) -> bool:
        if not self._late:
            return True
        flat, _ = self._model.flat_parameters()
        lo = (t.data_ptr() - flat.data_ptr()) // 4
        if lo < 0 or lo >= flat.numel():
            return True  # not a view of this model's flat parameter buffer
        hi = lo + t.numel()
        return all(not (off < hi and lo < off + n) for _, spans in self._late for off, n in spans)

    def _late_group(self):
        """(first flat element, [sets of data pointers]) of the weights that may be updated late: the large conv weights
        whose gradient the next step's lazy zero_grad does not touch, behind the encoder's parameters -- ONE group, waited for
        behind the encoder.  RAC_ADAM_LATE_GROUPS=1 groups them in the order the forward pass needs them
        (SVGConvModel.late_update_groups: prior, posterior, frame predictor + decoder; the recurrent core then waits chain by
        chain, ops.param_wait) so that two thirds of the pass run under the prior's and the posterior's gate GEMMs instead of
        under the encoder -- measured SLOWER (three same-box rounds: 23.07 ms no overlap, 22.75 one group, 23.41 staged groups):
        the gate GEMMs at M = 1024 stream their weights from HBM and lose more than the encoder does."""
        model = self._model
        lazy = getattr(model, "_lazy_params", None)
        if not self.overlap_next_forward or not lazy or not hasattr(model, "_encoder_extent"):
            return None
        lazy_ptrs = {p.data_ptr() for p in lazy}
        groups = getattr(model, "late_update_groups", None) if os.environ.get("RAC_ADAM_LATE_GROUPS", "0") == "1" else None
        sets, seen = [], set()
        for grp in (groups() if groups is not None else []):
            ptrs = frozenset(p.data_ptr() for p in grp if p.data_ptr() in lazy_ptrs and p.data_ptr() not in seen)
            seen |= ptrs
            if ptrs:
                sets.append(ptrs)
        rest = frozenset(lazy_ptrs - seen)
        if rest:
            sets.append(rest)
        return model._encoder_extent(), sets

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
        if ops._STALE:  # a lazy zero_grad whose step never reached finish_grads(): no stale gradient is ever applied
            ops.finish_grads()
        # one pass that also writes the next step's operand parts of the split-precision conv weights, when their parts
        # and maxima are current (every step after the first); plain Adam over the whole buffer otherwise
        self.wait_params()  # (a late update never outlives the forward pass that follows it; a caller without one: here)
        res = ops.fused_adam_step(flat, grad, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._steps,
                                  late=self._late_group())
        if res:
            if res is not True:
                self._late = list(res)
                ops.PARAM_GATE = self
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
        self.wait_params()
        if self._steps:
            for p, mv, vv in self._views():
                self.state[p] = {"step": torch.tensor(float(self._steps)), "exp_avg": mv, "exp_avg_sq": vv}
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self.wait_params()
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


def shard_plan(total: int, world: int, bucket_elems: int):
    """Buckets of the flat buffers for the sharded optimiser step: [(start, size)] in flat-buffer order (= the order the
    forward pass first uses the weights), every size a multiple of 4 * world so that the `world` slices of a bucket are
    equal and 16-byte aligned.  `total` must itself be such a multiple (SVGConvModel pads its flat buffers to 1024)."""
    q = 4 * world
    if total % q:
        raise ValueError("sharded optimiser step: %d flat elements do not cut into %d equal 16-byte-aligned slices "
                         "(the flat buffers are padded to 1024 elements: the world size must divide 256)" % (total, world))
    step = max(q, bucket_elems // q * q)
    return [(s, min(step, total - s)) for s in range(0, total, step)]


class ShardedAdam(FusedAdam):
    """Data-parallel optimiser step with the optimiser state and the update sharded over the ranks (ZeRO-1 on the flat
    buffers; `--ddp_shard_optimizer True`), instead of all-reduce + the same full Adam on every rank:

      backward   reduce-scatter (SUM) of each gradient bucket: rank r receives slice r of the bucket -- the bytes of HALF an
                 all-reduce stand between the last weight gradient and the optimiser (`ShardReducer`);
      step       1 / world scale and Adam on the owned slices only: 1 / world of the 6.7 GB Adam traffic, exp_avg /
                 exp_avg_sq exist only for the owned 1 / world of the parameters;
      next step  all-gather of the updated parameter buckets, asynchronous and in the order the forward pass needs the
                 weights; the model waits for the encoder's buckets before its first kernel and for the rest behind the
                 encoder's forward pass (`SVGConvModel._encode`: `wait_params(upto=...)`, then `wait_params()`), so the
                 ConvLSTM weights' 90 % of the bytes travel under the encoder; state_dict waits for everything.

    The conv weights' split-precision operand parts are refreshed lazily after the all-gather (ops._wp_refresh: one
    absmax + one fragment-split launch over all weights) -- the fused Adam + parts pass needs the whole updated weight.
    Same arithmetic per element as FusedAdam (`rac_adam_step` on slices), so a one-rank group reproduces it bit for bit."""

    @staticmethod
    def supports(model, world: int) -> bool:
        return model.flat_parameters()[0].numel() % (4 * world) == 0

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, bucket_mb: int = 64):
        super().__init__(model, lr=lr, betas=betas, eps=eps)
        if dist.is_available() and dist.is_initialized():
            shard_plan(model.flat_parameters()[0].numel(), dist.get_world_size(), 1)  # a clear error now, not in step 1
        self.bucket_elems = max(1, int(bucket_mb)) * (1 << 20) // 4
        self._pending = []   # async all-gather works of the last step's parameters
        self._adam = None    # tests on the CPU inject a torch Adam here; the product path is the HIP kernel
        self._ms = self._vs = None

    def plan(self):
        flat, _ = self._model.flat_parameters()
        world = dist.get_world_size()
        return shard_plan(flat.numel(), world, self.bucket_elems), world, dist.get_rank()

    def _shard_moments(self, n_own, device):
        if self._ms is None or self._ms.numel() != n_own or self._ms.device != device:
            self._ms = torch.zeros(n_own, device=device, dtype=torch.float32)
            self._vs = torch.zeros(n_own, device=device, dtype=torch.float32)
        return self._ms, self._vs

    def wait_params(self, upto: int = None):
        """Make the current stream (gloo: the host) wait for the parameter all-gather: every bucket, or the buckets that
        hold flat elements [0, upto) -- the collectives complete in issue order, so the rest stays in flight behind the
        kernels enqueued next (the encoder's forward pass runs under the ConvLSTM weights' all-gather)."""
        keep = []
        for start, size, work in self._pending:
            if upto is None or start < upto:
                work.wait()
            else:
                keep.append((start, size, work))
        self._pending = keep
        if not keep and ops.PARAM_GATE is self:
            ops.PARAM_GATE = None

    def wait_for(self, t: torch.Tensor):
        """(the all-gather's buckets complete in issue order: waiting for a parameter is waiting for the buckets up to its end)"""
        if not self.ready(t):
            flat, _ = self._model.flat_parameters()
            self.wait_params(upto=(t.data_ptr() - flat.data_ptr()) // 4 + t.numel())

    def ready(self, t: torch.Tensor) -> bool:
        """Have the buckets overlapping the parameter memory `t` been waited for?"""
        if not self._pending:
            return True
        flat, _ = self._model.flat_parameters()
        lo = (t.data_ptr() - flat.data_ptr()) // 4
        if lo < 0 or lo >= flat.numel():
            return True  # not a view of the flat parameter buffer
        hi = lo + t.numel()
        return all(not (start < hi and lo < start + size) for start, size, _ in self._pending)

    @torch.no_grad()
    def step(self, closure=None):
        """Expects the gradient buckets reduce-scattered (ShardReducer.finish): slice `rank` of every bucket of the flat
        gradient holds the SUM over ranks."""
        flat, grad = self._model.flat_parameters()
        buckets, world, rank = self.plan()
        g = self.param_groups[0]
        self._steps += 1
        self.wait_params()  # (normally long done: the model waited before it read the parameters)
        if ops._STALE:
            ops.finish_grads()
        n_own = sum(size // world for _, size in buckets)
        ms, vs = self._shard_moments(n_own, flat.device)
        pos = 0
        inv = 1.0 / world
        for start, size in buckets:
            n = size // world
            lo = start + rank * n
            p_s, g_s = flat[lo:lo + n], grad[lo:lo + n]
            if world > 1:
                g_s.mul_(inv)
            if self._adam is not None:
                self._adam(p_s, g_s, ms[pos:pos + n], vs[pos:pos + n], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                           self._steps)
            else:
                if not flat.is_cuda:
                    raise _lib.RacError("ShardedAdam runs on the GPU only")
                _lib.call("rac_adam_step", p_s.data_ptr(), g_s.data_ptr(), ms[pos:pos + n].data_ptr(),
                          vs[pos:pos + n].data_ptr(), n, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                          float(g["eps"]), self._steps, _lib.stream_ptr())
            pos += n
        # the updated slices travel to every rank, bucket by bucket in forward order, while the host moves on
        for start, size in buckets:
            n = size // world
            work = dist.all_gather_into_tensor(flat[start:start + size], flat[start + rank * n:start + (rank + 1) * n],
                                               async_op=True)
            self._pending.append((start, size, work))
        ops.PARAM_EPOCH += 1  # caches derived from the parameters (padded copies, operand parts) are stale
        ops.PARAM_GATE = self  # ... and are refreshed bucket by bucket as the all-gather is waited for (ops._wp_refresh)

    # torch.optim.Adam-compatible state: the full moments are assembled from all ranks' slices
    def _moments(self):
        flat, _ = self._model.flat_parameters()
        buckets, world, rank = self.plan()
        n_own = sum(size // world for _, size in buckets)
        ms, vs = self._shard_moments(n_own, flat.device)
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        pos = 0
        for start, size in buckets:
            n = size // world
            dist.all_gather_into_tensor(m[start:start + size], ms[pos:pos + n].contiguous())
            dist.all_gather_into_tensor(v[start:start + size], vs[pos:pos + n].contiguous())
            pos += n
        return m, v

    def state_dict(self):
        self.wait_params()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        torch.optim.Optimizer.load_state_dict(self, state_dict)
        flat, _ = self._model.flat_parameters()
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        steps = 0
        for p in self.param_groups[0]["params"]:
            st = self.state.get(p)
            if st:
                torch.as_strided(m, p.shape, p.stride(), p._rac_off).copy_(st["exp_avg"])
                torch.as_strided(v, p.shape, p.stride(), p._rac_off).copy_(st["exp_avg_sq"])
                steps = max(steps, int(st["step"]))
        buckets, world, rank = self.plan()
        ms, vs = self._shard_moments(sum(size // world for _, size in buckets), flat.device)
        pos = 0
        for start, size in buckets:
            n = size // world
            ms[pos:pos + n].copy_(m[start + rank * n:start + (rank + 1) * n])
            vs[pos:pos + n].copy_(v[start + rank * n:start + (rank + 1) * n])
            pos += n
        self._steps = steps
        self.state.clear()

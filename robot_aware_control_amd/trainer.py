"""PredictionTrainer: the SVG train step on librac_hip.so, API-compatible with the reference
`src/prediction/trainer.py` (`PredictionTrainer(config)`, `_train_step`, `_train_video`, `train`,
checkpoint format `{"model","optimizer","step"}`, ckpt_{step}.pt discovery).

MI355X-first differences:
  * no host sync inside the time loop: the 4 `.item()` calls per time step of the reference
    (trainer.py:433-458) become one device->host copy per train step;
  * zero_robot_region of the input frame is fused into the encoder's input packing;
  * parameter gradients accumulate in place in one flat buffer; the DDP gradient all-reduce
    (RCCL over xGMI, one process per GPU) runs on slices of that buffer, Adam is one launch.
"""
from __future__ import annotations

import os
from collections import defaultdict
from glob import glob
from math import floor

import numpy as np
import torch
import torch.distributed as dist

from . import ops, parallel_env
from .model import SVGConvModel
from .optim import FusedAdam, ShardedAdam

# teacher-forced windows run the encoder / decoder once over all time steps (RAC_SEQUENCE_PATH=0: step by step)
SEQUENCE_PATH = os.environ.get("RAC_SEQUENCE_PATH", "1") == "1"
STEP_HIGH_PRIORITY = os.environ.get("RAC_STEP_HIGH_PRIORITY", "0") == "1"


def _dist_on() -> bool:
    return parallel_env.active()


def allreduce_flat_grad(flat_grad: torch.Tensor, bucket_mb: int = 64):
    """Mean of the flat gradient over all ranks: bucketed async all-reduce (RCCL; each bucket is a
    contiguous slice, so there is no pack/unpack copy), then one scale."""
    if not _dist_on():
        return
    red = GradReducer(flat_grad, bucket_mb)
    red.finish()


class GradReducer:
    """Data-parallel mean of the flat gradient buffer, overlapped with the tail of backward.

    The ConvLSTM gate weights are 90 % of the parameters (214 M of 238.6 M at g512) and their gradients are the LAST
    thing backward produces (one time-batched wgrad launch per weight when `ops.deferred_wgrad()` exits).  `ready(p)`
    is called right after a weight's launch is enqueued: the all-reduce of that weight's slice starts on RCCL's
    stream while the next weight's wgrad computes.  `finish()` reduces whatever has not been reduced yet (vgg
    stack, input convs, heads, biases: 10 %), waits for everything and applies the 1/world scale once."""

    def __init__(self, flat_grad: torch.Tensor, bucket_mb: int = 64):
        self.flat = flat_grad
        self.step = max(1, bucket_mb * (1 << 20) // 4)
        self.works = []
        self.done = []  # reduced [start, end) element ranges

    def _reduce(self, start: int, end: int):
        for s in range(start, end, self.step):
            e = min(end, s + self.step)
            self.works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def ready(self, param: torch.Tensor):
        g = param.grad
        if g is None or g.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
            return  # not a view of the flat buffer: left to finish()
        start = g.storage_offset() - self.flat.storage_offset()
        self._reduce(start, start + g.numel())
        self.done.append((start, start + g.numel()))

    def finish(self):
        pos = 0
        for start, end in sorted(self.done):
            if start > pos:
                self._reduce(pos, start)
            pos = max(pos, end)
        if pos < self.flat.numel():
            self._reduce(pos, self.flat.numel())
        for w in self.works:
            w.wait()
        self.flat.mul_(1.0 / dist.get_world_size())


class ShardReducer:
    """The sharded optimiser's half of the exchange (optim.ShardedAdam): a reduce-scatter (SUM) per bucket of the flat
    gradient, issued as soon as every weight gradient inside the bucket is final -- the ConvLSTM weights' buckets right
    after their time-batched wgrad launch (`ready`), the rest in `finish()`.  Rank r's slice of each bucket is reduced
    IN PLACE (output = that slice of the input bucket)."""

    def __init__(self, flat_grad: torch.Tensor, buckets, world: int, rank: int):
        self.flat, self.buckets, self.world, self.rank = flat_grad, buckets, world, rank
        self.covered = [0] * len(buckets)
        self.issued = [False] * len(buckets)
        self.works = []

    def _issue(self, b: int):
        start, size = self.buckets[b]
        n = size // self.world
        bucket = self.flat[start:start + size]
        self.works.append(dist.reduce_scatter_tensor(bucket[self.rank * n:(self.rank + 1) * n], bucket,
                                                     op=dist.ReduceOp.SUM, async_op=True))
        self.issued[b] = True

    def ready(self, param: torch.Tensor):
        g = param.grad
        if g is None or g.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
            return
        lo = g.storage_offset() - self.flat.storage_offset()
        hi = lo + g.numel()
        for b, (start, size) in enumerate(self.buckets):
            ov = min(hi, start + size) - max(lo, start)
            if ov > 0 and lo <= start and start + size <= hi and not self.issued[b]:
                self._issue(b)  # the bucket lies inside this weight's gradient: final now

    def finish(self):
        for b in range(len(self.buckets)):
            if not self.issued[b]:
                self._issue(b)
        for w in self.works:
            w.wait()


class PredictionTrainer(object):
    """Video prediction training (reference trainer.py:53-897, hot path only)."""

    def __init__(self, config, logger=None):
        self._config = config
        if not torch.cuda.is_available():
            raise ops._lib.RacError("PredictionTrainer needs an MI355X (no CPU fallback)")
        local = int(os.environ.get("LOCAL_RANK", 0))
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        self._device = config.device = device
        self._logger = logger
        self.robot_model = None  # finetune_* experiments: predict_batch(batch) -> (states, masks)
        self._init_models(config)
        self._scheduled_sampling = config.scheduled_sampling
        self._step = 0
        self._plot_rng = np.random.RandomState(self._config.seed)
        self._video_sample_rng = np.random.RandomState(self._config.seed)
        self._grad_seeds = {}
        self._loss_host = None  # pinned staging buffer of the per-step loss readback
        self.phase_events = None  # set to [] to record (name, cuda event) marks of every train step (bench.py)
        # the model / optimiser object graph is static from here on: keep the cyclic GC's full collections from
        # walking it (measured: one 33 ms host stall every ~10 train steps at cfg2, during which the GPU drains)
        if os.environ.get("RAC_GC_FREEZE", "0") == "1":  # opt-in (bench.py, the CLI): process-global and irreversible
            import gc
            gc.collect()
            gc.freeze()
        self._wandb = None
        if getattr(config, "wandb", False):
            import wandb  # only when asked for (reference trainer.py:70-84)
            wandb.init(resume=config.jobname, project=config.wandb_project, config=config, dir=config.log_dir,
                       entity=config.wandb_entity, group=config.wandb_group, job_type=config.wandb_job_type)
            self._wandb = wandb

    # ------------------------------------------------------------------ setup
    def _init_models(self, cf):
        if cf.model != "svg":  # trainer.py:99-107: det / copy / cdna_det are outside the accelerated path
            raise ValueError(f"{cf.model}: only --model svg is built on the HIP path")
        self.model = SVGConvModel(cf).to(self._device)
        if _dist_on():  # identical initial weights on every rank
            dist.broadcast(self.model.flat_parameters()[0], src=0)
        if cf.optimizer != "adam":
            raise ValueError("Unknown optimizer on the HIP path: %s" % cf.optimizer)
        shard = getattr(cf, "ddp_shard_optimizer", False) and _dist_on()
        if shard and not ShardedAdam.supports(self.model, dist.get_world_size()):
            # equal 16-byte-aligned slices need the flat buffers (padded to 1024 elements) to divide by 4 * world
            import warnings
            warnings.warn("--ddp_shard_optimizer needs a world size that divides 256 (got %d): "
                          "using all-reduce + the full Adam step instead" % dist.get_world_size())
            shard = False
        if shard:
            self.optimizer = ShardedAdam(self.model, lr=cf.lr, betas=(cf.beta1, 0.999),
                                         bucket_mb=getattr(cf, "ddp_bucket_mb", 64))
        else:
            self.optimizer = FusedAdam(self.model, lr=cf.lr, betas=(cf.beta1, 0.999))

    def _schedule_prob(self):
        """Probability of feeding ground truth (trainer.py:132-140)."""
        k = self._config.scheduled_sampling_k
        use_truth = k / (k + np.exp(self._step / k))
        return [use_truth, 1 - use_truth]

    def _use_true_token(self):
        """Scheduled sampling coin (trainer.py:142-147)."""
        if not self._scheduled_sampling:
            return True
        return np.random.choice([True, False], p=self._schedule_prob())

    def _loss_kind(self):
        kind = self._config.reconstruction_loss
        if kind not in ops.LOSS_KINDS:
            raise NotImplementedError(f"{kind}")
        return ops.LOSS_KINDS[kind]

    def _recon_loss(self, prediction, target, mask=None, batch_weight=None):
        """trainer.py:149-161; returns the (3,) [loss, robot_mse, world_mse] tensor of the fused kernel."""
        cf = self._config
        rw = cf.robot_pixel_weight if "dontcare" in cf.reconstruction_loss else 0.0
        bw = batch_weight if cf.reconstruction_loss in ("l1", "dontcare_l1") else None
        return ops.ReconLoss.apply(prediction, target, mask, bw, self._loss_kind(), rw)

    def _mark(self, name):
        if self.phase_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.phase_events.append((name, e))

    def _seed(self, value: float, n: int = 1, first_only: bool = False):
        key = (value, n, first_only)
        if key not in self._grad_seeds:
            t = torch.zeros(n, device=self._device)
            if first_only:
                t[0] = value
            else:
                t.fill_(value)
            self._grad_seeds[key] = t
        return self._grad_seeds[key]

    # ------------------------------------------------------------- train step
    def _train_video(self, data):
        """Slice a video into n_past+n_future windows and train on each (trainer.py:259-324)."""
        cf = self._config
        x = data["images"]
        T = len(x)
        window = cf.n_past + cf.n_future
        self.steps_per_train_video = floor(T / window)
        all_losses = defaultdict(float)
        for i in range(floor(T / window)):
            if cf.random_snippet:
                s = self._video_sample_rng.randint(0, (T - window) + 1)
                e = s + window
            else:
                s, e = i * window, (i + 1) * window
            batch = {"images": x[s:e], "states": data["states"][s:e], "actions": data["actions"][s:e - 1],
                     "masks": data["masks"][s:e], "robot": data["robot"], "folder": data.get("folder")}
            if "qpos" in data:
                batch["qpos"] = data["qpos"][s:e]
            if getattr(cf, "model_use_heatmap", False):
                batch["heatmaps"] = data["heatmaps"][s:e]
            if cf.load_movement_info:
                batch["high_movement"] = data["high_movement"]
            if "finetune" in cf.experiment and (cf.model_use_mask or cf.model_use_robot_state):
                # finetune_*: states and masks of the window come from a robot model (trainer.py:294-319)
                if getattr(cf, "preprocess_action", "raw") != "raw":
                    batch["raw_actions"], batch["raw_states"] = data["raw_actions"][s:e - 1], data["raw_states"][s:e]
                    batch["raw_low"], batch["raw_high"] = data["raw_low"], data["raw_high"]
                batch["low"], batch["high"] = data["low"], data["high"]
                if getattr(self, "robot_model", None) is None:
                    raise NotImplementedError(
                        "finetune_* windows take their robot states and masks from an analytical / learned robot model "
                        "(trainer.py:294-319): set `trainer.robot_model` to an object with predict_batch(batch) -> "
                        "(states, masks) -- the reference's LocobotAnalyticalModel, or robot_atlas.AtlasRobotModel")
                out = self.robot_model.predict_batch(batch)
                if getattr(cf, "model_use_heatmap", False):
                    batch["states"], batch["masks"], batch["heatmaps"] = out
                else:
                    batch["states"], batch["masks"] = out
            losses = self._train_step(batch)
            for k, v in losses.items():
                all_losses[k] += v / floor(T / window)
        return all_losses

    def _train_step(self, data, use_truth=None):
        """Forward and backward pass + optimiser step (trainer.py:326-465).  Returns the loss dict.
        `use_truth[i]` overrides the scheduled-sampling coin at time index i (parity tests)."""
        if STEP_HIGH_PRIORITY:
            # (experiment) the whole step on a HIGH-priority stream: when the side stream's weight-gradient workgroups retire,
            # the step's own small kernels get the freed slots first
            if getattr(self, "_hp_stream", None) is None:
                self._hp_stream = torch.cuda.Stream(device=self._device, priority=-1)
            cur = torch.cuda.current_stream()
            self._hp_stream.wait_stream(cur)
            with torch.cuda.stream(self._hp_stream):
                out = self._train_step_body(data, use_truth)
            cur.wait_stream(self._hp_stream)
            return out
        return self._train_step_body(data, use_truth)

    def _train_step_body(self, data, use_truth=None):
        cf = self._config
        dev = self._device
        f32 = torch.float32
        x = data["images"].to(dev, f32)
        states = data["states"].to(dev, f32)
        ac = data["actions"].to(dev, f32)
        mask = data["masks"].to(dev, f32)
        heatmaps = data["heatmaps"].to(dev, f32) if getattr(cf, "model_use_heatmap", False) else None
        robot_name = np.array(data["robot"])
        all_robots = sorted(set(robot_name))
        batch_weight = None
        if cf.load_movement_info:
            mv = data["high_movement"].to(dev)
            batch_weight = (cf.movement_weight * mv).to(f32)
            batch_weight[~mv.bool()] = 1.0

        # (a sharded optimiser's parameter all-gather of the previous step is waited for inside the model: the encoder's
        # buckets before its first kernel, the rest behind the encoder's forward pass -- SVGConvModel._encode)
        ops.begin_step(x.device)
        self._mark("start")
        # (lazy: the large conv weights' gradients are written, not added to zeros, by their one launch per step)
        self.model.zero_grad(lazy=ops.LAZY_ZERO_GRAD)
        bs = min(cf.batch_size, x.shape[1])
        self.model.init_hidden(bs)
        dontcare = "dontcare" in cf.reconstruction_loss or cf.black_robot_input
        roots, seeds, log = [], [], []  # autograd roots, their incoming grads, (name, tensor, index) for the readback
        n_steps = cf.n_past + cf.n_future - 1

        def add_losses(x_pred, i, mu, logvar, mu_p, logvar_p, kl_here=True):
            x_i, m_i = x[i], mask[i]
            rec = self._recon_loss(x_pred, x_i.contiguous(), m_i.contiguous(), batch_weight)
            roots.append(rec)
            seeds.append(self._seed(1.0, 3, first_only=True))
            log.extend([("recon_loss", rec, 0), ("robot_loss", rec, 1), ("world_loss", rec, 2)])
            if len(all_robots) > 1:  # per-robot logging metrics (trainer.py:442-452)
                with torch.no_grad():
                    for r in all_robots:
                        idx = torch.from_numpy(np.nonzero(robot_name == r)[0]).to(dev)
                        sub = ops.ReconLoss.apply(x_pred.detach()[idx].contiguous(), x_i[idx].contiguous(),
                                                  m_i[idx].contiguous(), None, 0, 0.0)
                        log.extend([(f"{r}_robot_loss", sub, 1), (f"{r}_world_loss", sub, 2)])
            if kl_here:
                add_kl(mu, logvar, mu_p, logvar_p)

        def add_kl(mu, logvar, mu_p, logvar_p):
            kl = ops.KLLoss.apply(mu, logvar, mu_p, logvar_p, bs)
            roots.append(kl)
            seeds.append(self._seed(float(cf.beta)))
            log.append(("kld", kl, 0))

        # scheduled-sampling coins of the window, drawn in the reference's order (nothing else uses np.random here)
        truths = [True] + [(self._use_true_token() if use_truth is None else bool(use_truth[i]))
                           for i in range(2, n_steps + 1)]
        H, W = x.shape[-2], x.shape[-1]
        sequence_taken = False
        if SEQUENCE_PATH and all(truths) and self.model.sequence_ok(bs, H, W) and x.shape[1] == bs:
            sequence_taken = True
            # every input frame is ground truth: encoder and decoder run once over the whole window
            T = n_steps
            m_all = torch.cat([mask[:T], mask[1:T + 1]], 2) if cf.model_use_future_mask else mask[:T]
            hm_all = None
            if heatmaps is not None:
                hm_all = torch.cat([heatmaps[:T], heatmaps[1:T + 1]], 2) if cf.model_use_future_heatmap else heatmaps[:T]
            robots = [((states[t], states[t + 1]) if cf.model_use_future_robot_state else states[t]) for t in range(T)]
            x4, mus, logvars, mu_ps, logvar_ps = self.model.forward_sequence_maps(
                x[:T], m_all, robots, hm_all, [ac[t] for t in range(T)], [states[t + 1] for t in range(T)],
                mask[:T] if dontcare else None)
            x_pred_all = ops.Composite.apply(x4, x[:T].reshape((T * bs,) + tuple(x.shape[2:])).contiguous())
            x_preds = x_pred_all.view((T, bs) + tuple(x_pred_all.shape[1:])).unbind(0)
            batched = self.model.sequence_batched
            if batched is not None and len(all_robots) == 1:
                # sum_t recon_t = sum_t mean_b f(t, b) = T * mean over ALL T*B samples: one launch over the window (every
                # loss kind normalises per sample, losses.py:11-50), seeded with T; the logged terms are scaled back
                bw_all = None if batch_weight is None else batch_weight.repeat(T)
                flatw = lambda t_: t_[1:T + 1].reshape((T * bs,) + tuple(t_.shape[2:])).contiguous()
                rec = self._recon_loss(x_pred_all, flatw(x), flatw(mask), bw_all)
                roots.append(rec)
                seeds.append(self._seed(float(T), 3, first_only=True))
                log.extend([("recon_loss", rec, 0, T), ("robot_loss", rec, 1, T), ("world_loss", rec, 2, T)])
            else:
                for t in range(T):
                    add_losses(x_preds[t], t + 1, mus[t], logvars[t], mu_ps[t], logvar_ps[t], kl_here=batched is None)
            if batched is not None:
                # sum_t KL_t: every term is a sum over its elements / bs (losses.py:97-106), so the window's KL is ONE
                # launch over all T*B samples' elements (and one backward launch writing the batched gradients)
                add_kl(*batched)
        else:
            x_pred = None
            skip = None
            for i in range(1, n_steps + 1):
                x_j = x[i - 1] if truths[i - 1] else x_pred  # scheduled sampling: gradients flow through the fed-back frame
                m_j, r_j, a_j = mask[i - 1], states[i - 1], ac[i - 1]
                m_i, r_i = mask[i], states[i]
                if cf.last_frame_skip:
                    skip = None
                m_in = torch.cat([m_j, m_i], 1) if cf.model_use_future_mask else m_j
                r_in = (r_j, r_i) if cf.model_use_future_robot_state else r_j
                hm_in = None
                if heatmaps is not None:
                    hm_in = (torch.cat([heatmaps[i - 1], heatmaps[i]], 1) if cf.model_use_future_heatmap
                             else heatmaps[i - 1])
                x4, curr_skip, mu, logvar, mu_p, logvar_p = self.model.forward_maps(
                    x_j, m_in, r_in, hm_in, a_j, True, r_i, skip, zero_mask=m_j if dontcare else None)
                x_pred = ops.Composite.apply(x4, x_j.contiguous())  # un-blacked x_j (trainer.py:406-407)
                if i <= cf.n_past:
                    skip = curr_skip
                add_losses(x_pred, i, mu, logvar, mu_p, logvar_p)
        # the step's loss scalars are final once the forward pass is enqueued: start their device->host copy now (into
        # pinned memory) and collect it after the optimiser step is enqueued, so that the host never waits for the
        # backward pass or Adam (the reference likewise reads its losses before loss.backward(), trainer.py:433-458)
        self._mark("forward")
        with torch.no_grad():
            vals_dev = torch.stack([e[1].detach()[e[2]] for e in log])
        if self._loss_host is None or self._loss_host.numel() < vals_dev.numel():
            self._loss_host = torch.empty(max(64, vals_dev.numel()), dtype=vals_dev.dtype).pin_memory()
        vals_host = self._loss_host[:vals_dev.numel()]
        vals_host.copy_(vals_dev, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        # loss = sum_t recon_t + beta * sum_t kl_t (trainer.py:459): seed each term's gradient directly
        reducer = None
        if _dist_on():
            if isinstance(self.optimizer, ShardedAdam):
                reducer = ShardReducer(self.model.flat_parameters()[1], *self.optimizer.plan())
            else:
                reducer = GradReducer(self.model.flat_parameters()[1], getattr(cf, "ddp_bucket_mb", 64))
        # ConvLSTM weight gradients: one time-batched launch per weight, each followed by its slice's all-reduce.  In a window
        # that went step by step (scheduled sampling, GroupNorm cells) a weight's launch starts -- on the side stream -- the
        # moment its last step's operands exist (the backward pass reaches the window's first step last), under the rest of
        # that step's backward pass; the hand-scheduled core launches its chains' gradients itself
        stepped = not (sequence_taken and self.model.used_recurrent_core)
        try:
            fl = os.environ.get("RAC_SCHED_FLUSH")  # (experiments: "4,5" = flush points; default: the window's step count)
            points = tuple(int(v) for v in fl.split(",")) if fl else n_steps
            with ops.deferred_wgrad(on_ready=reducer.ready if reducer is not None else None,
                                    flush_after=points if stepped else None,
                                    vgg_steps=not sequence_taken):
                torch.autograd.backward(roots, seeds)
                self._mark("backward")
        finally:
            # a weight no launch wrote this step (none in the standard configurations) gets its zeros now -- also when
            # backward raised: a caller that catches the exception must not find last step's values in a .grad
            ops.finish_grads()
        self._mark("weight_grads")
        if reducer is not None:
            reducer.finish()
            self._mark("allreduce_exposed")
        self.optimizer.step()
        self._mark("adam")
        self.model.sequence_batched = None  # (graph references of this step)

        copied.synchronize()  # the one host wait of the step (normally already satisfied)
        vals = vals_host.tolist()
        losses = defaultdict(float)
        for e, v in zip(log, vals):
            losses[e[0]] += v * (e[3] if len(e) > 3 else 1)
        for k in losses:
            losses[k] = losses[k] / cf.n_future
        return losses

    # ------------------------------------------------------------------- eval
    def _compute_epoch_metrics(self, data_loader, name):
        """Average the per-video eval metrics over a loader (trainer.py:467-488)."""
        from .data import process_batch
        losses = defaultdict(list)
        for data in data_loader:
            data = process_batch(data, self._device)
            info = self._eval_video(data, autoregressive=True)
            for k, v in info.items():
                losses[k].append(v)
        return {f"{name}/{k}": float(np.mean(v)) for k, v in losses.items()}

    def _eval_video(self, data, autoregressive=False):
        """Evaluate a whole video in n_eval windows (trainer.py:490-564); ground-truth masks drive the rollout."""
        cf = self._config
        finetune = "finetune" in cf.experiment and (cf.model_use_mask or cf.model_use_robot_state)
        if finetune and getattr(self, "robot_model", None) is None:
            raise NotImplementedError(
                "finetune_* evaluation rolls out on the robot model's states and masks (trainer.py:520-543): set "
                "`trainer.robot_model` (the reference's LocobotAnalyticalModel, or robot_atlas.AtlasRobotModel)")
        x = data["images"]
        T = len(x)
        window = cf.n_eval
        total = defaultdict(float)
        for i in range(floor(T / window)):
            s, e = i * window, (i + 1) * window
            batch = {"images": x[s:e], "states": data["states"][s:e], "actions": data["actions"][s:e - 1],
                     "masks": data["masks"][s:e], "pred_masks": data["masks"][s:e], "robot": data["robot"]}
            if getattr(cf, "model_use_heatmap", False):
                batch["heatmaps"] = data["heatmaps"][s:e]
            if finetune:
                # predicted states / masks drive the rollout, the true masks score the world error (trainer.py:520-547)
                for k in ("qpos",):
                    if k in data:
                        batch[k] = data[k][s:e]
                if getattr(cf, "preprocess_action", "raw") != "raw":
                    batch["raw_actions"], batch["raw_states"] = data["raw_actions"][s:e - 1], data["raw_states"][s:e]
                    batch["raw_low"], batch["raw_high"] = data["raw_low"], data["raw_high"]
                batch["low"], batch["high"] = data["low"], data["high"]
                out = self.robot_model.predict_batch(batch)
                if getattr(cf, "model_use_heatmap", False):
                    batch["states"], batch["pred_masks"], batch["heatmaps"] = out
                else:
                    batch["states"], batch["pred_masks"] = out
            for k, v in self._eval_step(batch, autoregressive).items():
                total[k] += v
        for k in total:
            total[k] /= floor(T / window)
        return total

    @torch.no_grad()
    def _eval_step(self, data, autoregressive=False):
        """Evaluate one n_eval snippet (trainer.py:566-734): prior-driven prediction (`force_use_prior`), optional
        autoregressive feeding, recon / robot / world losses, PSNR, SSIM, KL.  One host sync at the end."""
        from .metrics import masked_psnr_ssim
        cf = self._config
        dev, f32 = self._device, torch.float32
        x = data["images"].to(dev, f32)
        states = data["states"].to(dev, f32)
        ac = data["actions"].to(dev, f32)
        true_masks = data["masks"].to(dev, f32)
        masks = data["pred_masks"].to(dev, f32)
        heatmaps = data["heatmaps"].to(dev, f32) if getattr(cf, "model_use_heatmap", False) else None
        robot_name = np.array(data["robot"])
        all_robots = sorted(set(robot_name))
        bs = min(cf.test_batch_size, x.shape[1])
        self.model.init_hidden(bs)
        prefix = "autoreg" if autoregressive else "1step"
        dontcare = "dontcare" in cf.reconstruction_loss or cf.black_robot_input
        log, klog = [], []
        x_pred = skip = None
        for i in range(1, cf.n_eval):
            x_j = x_pred if (autoregressive and i > 1) else x[i - 1]
            m_j, r_j, a_j = masks[i - 1], states[i - 1], ac[i - 1]
            m_i, r_i = masks[i], states[i]
            if cf.last_frame_skip:
                skip = None
            m_in = torch.cat([m_j, m_i], 1) if cf.model_use_future_mask else m_j
            r_in = (r_j, r_i) if cf.model_use_future_robot_state else r_j
            hm_in = None
            if heatmaps is not None:
                hm_in = torch.cat([heatmaps[i - 1], heatmaps[i]], 1) if cf.model_use_future_heatmap else heatmaps[i - 1]
            x4, curr_skip, mu, logvar, mu_p, logvar_p = self.model.forward_maps(
                x_j, m_in, r_in, hm_in, a_j, True, r_i, skip, force_use_prior=True,
                zero_mask=m_j if dontcare else None)
            x_pred = ops.Composite.apply(x4, x_j.contiguous())
            if i <= cf.n_past:
                skip = curr_skip
            tm = true_masks[i].contiguous()
            rec = self._recon_loss(x_pred, x[i].contiguous(), tm)
            log += [(f"{prefix}_recon_loss", rec[0]), (f"{prefix}_robot_loss", rec[1]), (f"{prefix}_world_loss", rec[2])]
            p, s = masked_psnr_ssim(x_pred, x[i].contiguous(), tm)  # robot region blacked with the true mask
            p = p.mean()
            log += [(f"{prefix}_psnr", p), (f"{prefix}_ssim", s)]
            if autoregressive:
                klog += [(f"{i}_step_psnr", p), (f"{i}_step_ssim", s), (f"{i}_step_world_loss", rec[2])]
            if len(all_robots) > 1:
                for r in all_robots:
                    idx = torch.from_numpy(np.nonzero(robot_name == r)[0]).to(dev)
                    sub = ops.ReconLoss.apply(x_pred[idx].contiguous(), x[i][idx].contiguous(), tm[idx].contiguous(),
                                              None, 0, 0.0)
                    log += [(f"{prefix}_{r}_robot_loss", sub[1]), (f"{prefix}_{r}_world_loss", sub[2])]
            kl = ops.KLLoss.apply(mu, logvar, mu_p, logvar_p, bs)
            log.append((f"{prefix}_kld", kl[0]))
        vals = torch.stack([t.reshape(()) for _, t in log + klog]).cpu().tolist()
        losses = defaultdict(float)
        for (name, _), v in zip(log, vals[:len(log)]):
            losses[name] += v
        for k in losses:
            losses[k] = losses[k] / (cf.n_eval - 1)
        for (name, _), v in zip(klog, vals[len(log):]):
            losses[name] = v
        return losses

    # ------------------------------------------------------------------- plots
    @torch.no_grad()
    def plot(self, data, epoch, name, random_start=True, instance=None):
        """GIF of prior-driven autoregressive generations next to the ground truth (trainer.py:949-1147): per video one
        row [ground truth | 3 samples], one GIF frame per time step, written to `<plot_dir>/<name>_<epoch>.gif`.
        `data`: a time-first batch.  Returns the file name."""
        from PIL import Image
        cf = self._config
        dev, f32 = self._device, torch.float32
        b = min(data["images"].shape[1], 25)
        length = cf.n_past + cf.n_future if name in ("comparison", "train") else cf.n_eval
        total = data["images"].shape[0]
        starts = (self._plot_rng.randint(0, total - length + 1, size=b) if random_start else np.zeros(b, dtype=np.int64))

        def clip(key, shorten=0):  # every video's own [start, start + length) window, batch truncated to b
            t = data[key]
            return torch.stack([t[s:s + length - shorten, i] for i, s in enumerate(starts)], 1).to(dev, f32)
        x, states, ac, mask = clip("images"), clip("states"), clip("actions", 1), clip("masks")
        heatmaps = clip("heatmaps") if getattr(cf, "model_use_heatmap", False) else None
        if "finetune" in cf.experiment and (cf.model_use_mask or cf.model_use_robot_state):
            if getattr(self, "robot_model", None) is None:
                raise NotImplementedError("finetune_* plots roll out on trainer.robot_model's states and masks")
            batch = {"states": states, "actions": ac, "masks": mask, "qpos": clip("qpos"), "low": data["low"][:b],
                     "high": data["high"][:b]}
            if getattr(cf, "preprocess_action", "raw") != "raw":
                batch.update(raw_low=data["raw_low"][:b], raw_high=data["raw_high"][:b],
                             raw_actions=clip("raw_actions", 1), raw_states=clip("raw_states"))
            states, mask = self.robot_model.predict_batch(batch)[:2]
        dontcare = "dontcare" in cf.reconstruction_loss or cf.black_robot_input
        was_training = self.model.training
        self.model.eval()
        samples = []
        for _ in range(3):  # nsample of the svg model
            self.model.init_hidden(b)
            x_j = x[0]
            frames = [ops.ZeroRegion.apply(x_j.contiguous(), mask[0].contiguous()) if dontcare else x_j]
            skip = None
            for i in range(1, length):
                m_j, m_i = mask[i - 1], mask[i]
                if cf.last_frame_skip:
                    skip = None
                m_in = torch.cat([m_j, m_i], 1) if cf.model_use_future_mask else m_j
                r_in = (states[i - 1], states[i]) if cf.model_use_future_robot_state else states[i - 1]
                hm_in = None
                if heatmaps is not None:
                    hm_in = torch.cat([heatmaps[i - 1], heatmaps[i]], 1) if cf.model_use_future_heatmap else heatmaps[i - 1]
                x4, curr_skip = self.model.forward_maps(x_j, m_in, r_in, hm_in, ac[i - 1], False, None, skip,
                                                        zero_mask=m_j if dontcare else None)[:2]
                x_pred = ops.Composite.apply(x4, x_j.contiguous())
                if i <= cf.n_past:
                    skip = curr_skip
                x_j = x[i] if i < cf.n_past else x_pred
                frames.append(ops.ZeroRegion.apply(x_pred, m_i.contiguous()) if dontcare else x_j)
            samples.append(torch.stack(frames))  # (length, b, 3, H, W)
        self.model.train(was_training)
        # one GIF frame per time step: b rows of [ground truth | sample 0 | sample 1 | sample 2]
        grid = torch.stack([x] + samples, 2)                      # (length, b, 4, 3, H, W)
        grid = grid.permute(0, 1, 4, 2, 5, 3).reshape(length, b * x.shape[-2], 4 * x.shape[-1], 3)
        imgs = [Image.fromarray(f) for f in (grid.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()]
        plot_dir = getattr(cf, "plot_dir", None) or os.path.join(cf.log_dir, "plot")
        os.makedirs(plot_dir, exist_ok=True)
        stem = f"{name}_{epoch}" if instance is None else f"{name}_ep{epoch}_{instance}"
        fname = os.path.join(plot_dir, stem + ".gif")
        imgs[0].save(fname, save_all=True, append_images=imgs[1:], duration=250, loop=0)
        if self._wandb is not None:
            self._wandb.log({f"{name}/gifs": self._wandb.Video(fname, format="gif")}, step=self._step)
        return fname

    # ----------------------------------------------------------- outer loops
    def train(self, batch_generator=None, test_hook=None, test_loader=None, transfer_loader=None):
        """Epoch loop with checkpoint and evaluation cadence (trainer.py:736-792).  `batch_generator` yields time-first
        batches in the layout of `process_batch` (robonet_dataset.py:434-451); by default the loaders come from
        `_setup_data`.  Every `eval_interval` epochs the model is evaluated on the test loader
        (`_compute_epoch_metrics`, keys `test/*`) and, for `--experiment train_robonet` with locobot data under the data
        root, on the zero-shot transfer loader (`transfer/*`) as the reference does; with `--plot True` (this repo's
        switch, default off) a GIF of generations is written per epoch / evaluation (`plot`)."""
        cf = self._config
        self._step = self._load_checkpoint(cf.dynamics_model_ckpt)
        # this loop owns every reader of the parameters between two steps (forward passes and checkpoints, which wait):
        # the optimiser may finish the large weights' update under the next step's encoder (optim.FusedAdam)
        if os.environ.get("RAC_ADAM_OVERLAP", "1") == "1" and isinstance(self.optimizer, FusedAdam):
            self.optimizer.overlap_next_forward = True
        if batch_generator is None:
            batch_generator, test_loader = self._setup_data()
            transfer_loader = transfer_loader or getattr(self, "transfer_loader", None)
        plots = bool(getattr(cf, "plot", False)) and (not _dist_on() or dist.get_rank() == 0)

        def evaluate(loader, name, epoch):
            from .data import process_batch
            info_ = self._compute_epoch_metrics(loader, name)
            self.eval_history.append((epoch, info_))
            if self._wandb is not None:
                self._wandb.log(info_, step=self._step)
            if plots:
                self.plot(process_batch(next(iter(loader)), self._device), epoch, name)
        gen = batch_generator
        info = {}
        self.eval_history = []
        for epoch in range(cf.niter):
            self.model.train()
            for _ in range(cf.epoch_size):
                data = next(gen)
                info = self._train_video(data)
                if self._scheduled_sampling:
                    info["sample_schedule"] = self._schedule_prob()[0]
                self._step += self.steps_per_train_video
                if plots and _ == cf.epoch_size - 1:
                    self.plot(data, epoch, "train")
                if self._wandb is not None:
                    self._wandb.log({f"train/{k}": v for k, v in info.items()}, step=self._step)
            if epoch % cf.checkpoint_interval == 0 and epoch > 0:
                self._save_checkpoint()
            if epoch % cf.eval_interval == 0:
                self.model.eval()
                if test_loader is not None:
                    evaluate(test_loader, "test", epoch)
                if transfer_loader is not None:
                    evaluate(transfer_loader, "transfer", epoch)
                if test_hook is not None:
                    test_hook(self, epoch)
        self._save_checkpoint()
        # nothing of this loop's optimiser stays installed behind it (ops.PARAM_GATE is process-global: it would pin the
        # trainer's 4 GB and be asked about the next model's parameters)
        self.optimizer.wait_params()
        if ops.PARAM_GATE is self.optimizer:
            ops.PARAM_GATE = None
        return info

    def _setup_data(self):
        """(infinite time-first batch generator, test loader).  `--data_root synthetic` feeds the deterministic
        generator of robot_aware_control_amd.synthetic; anything else goes through this package's RoboNet data path
        (data.py: same files, split and preprocessing as robonet_dataloaders.py:21-80, device prefetch)."""
        cf = self._config
        from . import data as D
        rank = dist.get_rank() if _dist_on() else 0
        if cf.data_root == "synthetic":
            from .synthetic import synth_video

            def gen():
                seed = cf.seed + 1000003 * rank  # every data-parallel rank trains on its own videos
                while True:
                    seed += 1
                    yield synth_video(seed, cf.video_length, cf.batch_size, cf.image_height, cf.image_width,
                                      cf.robot_dim, cf.action_dim)
            n_eval = getattr(cf, "n_eval", cf.n_past + cf.n_future)
            test = torch.utils.data.DataLoader(
                D.SyntheticVideoDataset(2 * getattr(cf, "test_batch_size", cf.batch_size), max(n_eval, cf.video_length),
                                        cf.image_height, cf.image_width, cf.robot_dim, cf.action_dim, seed=cf.seed + 7),
                batch_size=getattr(cf, "test_batch_size", cf.batch_size))
            return gen(), test
        train_loader, test_loader = D.create_loaders(cf)
        if cf.experiment == "train_robonet":  # zero-shot performance on the unseen locobot (trainer.py:899-913)
            self.transfer_loader = D.create_transfer_loader(cf)
        return D.get_batch(train_loader, self._device), test_loader

    def _save_checkpoint(self):
        """trainer.py:829-837.  Rank 0 writes the file; with the sharded optimiser EVERY rank first joins the collectives
        that assemble the full Adam moments (ShardedAdam.state_dict: all-gathers) and waits for the parameter all-gather of
        the last step, so that the parameters rank 0 clones are the updated ones."""
        opt_sd = None
        self.optimizer.wait_params()  # (parameters whose update is still in flight: FusedAdam's late group, an all-gather)
        if isinstance(self.optimizer, ShardedAdam):
            opt_sd = self.optimizer.state_dict()  # collective: before the rank check
        if _dist_on() and dist.get_rank() != 0:
            return
        os.makedirs(self._config.log_dir, exist_ok=True)
        path = os.path.join(self._config.log_dir, f"ckpt_{self._step}.pt")
        sd = {k: v.detach().clone().contiguous() if v.dim() != 4 else v.detach().clone()
              for k, v in self.model.state_dict().items()}
        if opt_sd is None:
            opt_sd = self.optimizer.state_dict()
        torch.save({"model": sd, "optimizer": opt_sd, "step": self._step}, path)
        return path

    def _load_checkpoint(self, ckpt_path=None):
        """Load a given checkpoint, else the newest ckpt_*.pt of log_dir (trainer.py:846-897)."""
        cf = self._config
        if ckpt_path is None:
            best, best_step = None, 0
            for f in sorted(glob(os.path.join(cf.log_dir, "*.pt"))):
                try:
                    num = int(os.path.basename(f).split(".")[0].rsplit("_", 1)[-1])
                except ValueError:
                    continue
                if num > best_step:
                    best, best_step = f, num
            if best is None:
                return 0
            ckpt_path = best
            finetune_reset = False
        else:
            finetune_reset = "finetune" in cf.experiment
        ckpt = torch.load(ckpt_path, map_location=self._device)
        self.model.load_state_dict(ckpt["model"])
        if finetune_reset:
            return 0
        self.optimizer.load_state_dict(ckpt["optimizer"])
        return ckpt["step"]

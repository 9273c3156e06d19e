"""TrajectorySampler: batched candidate rollouts through the frozen SVG model, on the GPU
(API of reference src/cem/trajectory_sampler.py:15-199).

Per (batch, step) the reference runs model.forward, two elementwise passes, a cost reduction and a
device->host copy (70 syncs per CEM iteration at N=1000).  Here the step tail -- compositing,
zero_robot_region, the image cost and its fp64 accumulation -- is one kernel and the per-candidate
sums stay on the device until the end: one D2H per call.

Multi-GPU: candidates are independent, so with torch.distributed initialised each rank rolls out
its contiguous slice and ONE all-gather (RCCL over xGMI) of the per-candidate fp64 costs precedes
elite selection.
"""
from __future__ import annotations

import os
import time as timer
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops, parallel_env
from .losses import RobotWorldCost
from .state import DemoGoalState, State


CEM_STREAMS = int(os.environ.get("RAC_CEM_STREAMS", "1"))  # parts of a candidate pass on their own streams (cfg.cem_streams)


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous, near-equal candidate slices; the tail ranks get the remainder-free part."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_costs(local: torch.Tensor, N: int, world: int, rank: int) -> np.ndarray:
    """All ranks' per-candidate fp64 cost sums -> one float64[N] on every rank (one all-gather; RCCL on the
    GPU path, any backend works).  Slices are padded to a common width because shards may differ by one."""
    if world == 1 and not (parallel_env.active() and dist.get_world_size() == 1):
        return local.cpu().numpy()  # (a one-rank group under RAC_DIST_FORCE=1 still runs the collective: RCCL rehearsal)
    width = (N + world - 1) // world
    send = torch.zeros(width, device=local.device, dtype=torch.float64)
    send[:local.numel()] = local
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send)
    out = []
    for r, part in enumerate(parts):
        lo, hi = shard_bounds(N, world, r)
        out.append(part[:hi - lo].cpu().numpy())
    return np.concatenate(out)


class TrajectorySampler(object):
    def __init__(self, cfg, model, cam_ext=None, franka_ik=None, wx250s_bot=None, push_height=None,
                 default_pitch=None, default_roll=None, robot_model=None) -> None:
        self.cfg = cfg
        self.model = model
        self.cost = RobotWorldCost(cfg)
        self.low = torch.tensor([[0.015, -0.3, 0.1, 0, 0]], dtype=torch.float32)   # trajectory_sampler.py:22-23
        self.high = torch.tensor([[0.55, 0.3, 0.4, 1, 1]], dtype=torch.float32)
        self.robot_model = robot_model
        self._refining = False  # inside the exact re-roll of the elites (no sharding, no nested refinement)
        self._streams = []      # extra HIP streams of a pass cut into `cem_streams` parts (generate_model_rollouts)
        self._robot_ctor = (cam_ext, franka_ik, wx250s_bot, push_height, default_pitch, default_roll)
        if os.environ.get("RAC_GC_FREEZE", "0") == "1":  # opt-in, see PredictionTrainer.__init__
            import gc
            gc.collect()
            gc.freeze()

    def _needs_robot(self):
        cfg = self.cfg
        return (cfg.model_use_robot_state or cfg.model_use_mask or cfg.black_robot_input
                or "dontcare" in cfg.reward_type)

    def _get_robot_model(self):
        """The analytical robot model (CPU IK + MuJoCo mask render, trajectory_sampler.py:26-33) is the
        reference's own; pass `robot_model=` or run inside the reference tree."""
        if self.robot_model is None:
            cam_ext, franka_ik, wx250s_bot, push_height, pitch, roll = self._robot_ctor
            if self.cfg.experiment == "control_franka":
                from src.dataset.franka.franka_model import FrankaAnalyticalModel
                self.robot_model = FrankaAnalyticalModel(self.cfg, franka_ik, cam_ext)
            else:
                from src.dataset.wx250s.wx250s_model import WX250sAnalyticalModel
                self.robot_model = WX250sAnalyticalModel(self.cfg, wx250s_bot, push_height, pitch, roll,
                                                         cam_ext=cam_ext)
        return self.robot_model

    def _robot_data(self, action_sequences, start, N, T):
        """The batch `predict_batch` is asked about: every candidate starts from the current robot state
        (trajectory_sampler.py:86-99)."""
        cfg = self.cfg
        states = torch.zeros((T + 1, N, 5), dtype=torch.float32)
        qpos = torch.zeros((T + 1, N, cfg.robot_joint_dim), dtype=torch.float32)
        start_state = torch.tensor(start.state)
        if cfg.experiment in ("control_franka", "control_wx250s"):
            from src.utils.camera_calibration import LOCO_FRANKA_DIFF, LOCO_WX250S_DIFF
            start_state[:2] = start_state[:2] + (LOCO_FRANKA_DIFF if cfg.experiment == "control_franka"
                                                 else LOCO_WX250S_DIFF)
        states[0, :] = (start_state - self.low) / (self.high - self.low)  # robonet_dataset.normalize
        qpos[0, :] = torch.tensor(start.qpos)
        return {"states": states, "qpos": qpos, "actions": action_sequences.permute(1, 0, 2),
                "low": self.low.repeat(N, 1), "high": self.high.repeat(N, 1)}

    def _predict_robot(self, action_sequences, start, N, T):
        """states (T+1,N,5) normalised, masks (T+1,N,1,H,W) for every candidate (trajectory_sampler.py:86-109)."""
        return self._get_robot_model().predict_batch(self._robot_data(action_sequences, start, N, T), thick=True)

    def _start_masks_shared(self, masks) -> bool:
        """Is row 0 of the robot model's masks (the start state's mask) the same for every candidate?  It is by
        construction of `_robot_data` (one start state for all candidates); answered WITHOUT a device readback on the
        planner's hot path: a model may say so itself (`shared_start_mask`: AtlasRobotModel does, from the host copy of
        the start states it was handed), host tensors (the analytical models return those) are compared on the host,
        and only an unknown model's device tensors cost one readback per call."""
        flag = getattr(self.robot_model, "shared_start_mask", None)
        if flag is not None:
            return bool(flag)
        m0 = masks[0]
        return bool((m0 == m0[:1]).all())  # CPU tensor: no device sync; device tensor: one readback per call

    def _refine_elites(self, sum_cost, action_sequences, start, goal):
        """`cfg.cem_exact_elites = M` with an atlas robot model that remembers the model it was rendered from
        (`AtlasRobotModel.exact`): the M best candidates of the atlas pass are rolled out again with EXACTLY rendered
        masks (the reference's own `predict_batch`, trajectory_sampler.py:86-109 -- M per-candidate renders instead of
        N) and their costs replaced, so the elite set is the one exact masks give whenever it lies inside the screened
        top M.  Every rank re-rolls the same M candidates (no collective when `sample_mean` or an `eps_source` makes the
        re-roll deterministic; otherwise rank 0's M costs are broadcast).  With `sample_mean=False` the re-rolled
        costs carry a fresh draw of the prior noise, like any second evaluation of a stochastic model would."""
        M = int(getattr(self.cfg, "cem_exact_elites", 0) or 0)
        exact = getattr(self.robot_model, "exact", None)
        if M <= 0 or exact is None or self._refining or not self._needs_robot():
            return sum_cost
        n = len(sum_cost)
        top = np.sort(np.argsort(-sum_cost, kind="stable")[:min(M, n)])
        atlas_model, self.robot_model, self._refining = self.robot_model, exact, True
        try:
            redo = self.generate_model_rollouts(action_sequences[:n][torch.from_numpy(top)].clone(), start, goal)
        finally:
            self.robot_model, self._refining = atlas_model, False
        redo_cost = np.asarray(redo["sum_cost"], dtype=np.float64)
        if not self.cfg.sample_mean and getattr(self.model, "eps_source", None) is None and parallel_env.active():
            # a stochastic re-roll draws its prior noise from each rank's own generator (the sharded pass consumed
            # different amounts of it): every rank takes rank 0's M costs, so that all ranks pick the same elites
            buf = torch.from_numpy(redo_cost.copy())
            if dist.get_backend() == "nccl":
                buf = buf.to(torch.device(self.cfg.device))
            dist.broadcast(buf, src=0)
            redo_cost = buf.cpu().numpy()
        sum_cost = sum_cost.copy()
        sum_cost[top] = redo_cost
        self.last_refined = top
        return sum_cost

    @torch.no_grad()
    def generate_model_rollouts(self, action_sequences, start: State, goal: DemoGoalState, opt_traj=None,
                                ret_obs=False, ret_step_cost=False, suppress_print=True):
        """Roll the candidate action sequences through the learned model and return
        {"sum_cost": float64[N], ...} exactly as the reference does."""
        cfg, model = self.cfg, self.model
        dev = torch.device(cfg.device)
        if dev.type != "cuda":
            raise _lib.RacError("generate_model_rollouts needs cfg.device on the GPU (no CPU fallback)")
        if opt_traj is not None:  # appended as candidate N+1 (trajectory_sampler.py:62-68)
            opt = torch.cat([opt_traj, torch.zeros((len(opt_traj), 3))], 1).unsqueeze(0)
            action_sequences = torch.cat([action_sequences, opt])
        N, T = action_sequences.shape[0], action_sequences.shape[1]
        H, W = cfg.image_height, cfg.image_width
        start_time = timer.time()

        goal_imgs = torch.stack([torch.from_numpy(g).permute(2, 0, 1).float() / 255 for g in goal.imgs]).to(dev)
        goal_masks = None
        if goal.masks is not None:
            goal_masks = torch.stack([torch.from_numpy(np.asarray(g)) for g in goal.masks]).to(dev).to(torch.uint8)
        dontcare_in = "dontcare" in cfg.reconstruction_loss or cfg.black_robot_input
        dontcare_cost = "dontcare" in cfg.reward_type
        kind = 1 if dontcare_cost else 0
        w_world = float(cfg.world_cost_weight)

        # ---- candidate sharding over ranks (one process per GPU) ----
        world, rank = 1, 0
        # the debug outputs (predicted frames / per-step costs of the elites, indexed over ALL candidates) are not
        # gathered: a call that asks for them rolls every candidate out on every rank
        if getattr(cfg, "cem_shard", True) and not (ret_obs or ret_step_cost) and not self._refining:
            world, rank = parallel_env.world_rank()
        lo, hi = shard_bounds(N, world, rank)
        n_local = hi - lo
        # the robot model is asked about THIS rank's candidates only (the reference asks about all N,
        # trajectory_sampler.py:86-109; a candidate's states and masks depend on nothing but its own actions): at cfg4 a
        # rank builds (15, 1000, 1, 64, 64) masks, not (15, 8000, ...).  `states` / `masks` are indexed shard-locally.
        states = masks = None
        shared_mask0 = True
        if self._needs_robot() and n_local > 0:
            states, masks = self._predict_robot(action_sequences[lo:hi], start, n_local, T)
            shared_mask0 = self._start_masks_shared(masks)
            states = states.to(dev, non_blocking=True)
            masks = masks.to(dev, torch.float32, non_blocking=True)
        per = cfg.candidates_batch_size
        nb = max(n_local // per, 1)
        sum_cost_dev = torch.zeros(max(n_local, 1), device=dev, dtype=torch.float64)
        all_obs = torch.zeros((N, T, 3, H, W)) if ret_obs else None
        step_cost = np.zeros((N, T)) if ret_step_cost else None
        start_img = (torch.from_numpy(start.img.copy()).permute(2, 0, 1).float() / 255).to(dev)
        actions_dev = action_sequences.to(dev, torch.float32)
        # step 0: every candidate sees the start frame (and the start mask, row 0 of the robot model's answer): the encoder
        # runs on one image.  A future mask is the candidate's own.  Decided once per call, without a device readback.
        shared0 = (getattr(cfg, "cem_shared_start", True) and not cfg.model_use_future_mask
                   and not getattr(cfg, "model_use_heatmap", False)
                   and (shared_mask0 or not (cfg.model_use_mask or dontcare_in)))

        # Candidates are independent and the frozen model's arithmetic does not depend on the batch (one operand scale per
        # image, K never split: DESIGN 3.1), so a pass may be cut into `cem_streams` parts that run on their own HIP streams
        # in lock-step -- the same bits, with one part's memory-bound kernels (first layer, output head, step tail) and launch
        # tails under the other's matrix-pipe kernels.  The first call of a sampler runs in one stream (it builds the cached
        # weight operands every stream reads afterwards).
        n_streams = int(getattr(cfg, "cem_streams", CEM_STREAMS))
        if (n_streams < 2 or ret_obs or ret_step_cost or not getattr(self, "_warm", False)
                or 64 * (T + 1) * max(per, 1) > ops._AMAX_SLOTS):
            n_streams = 1
        self._warm = True
        main = torch.cuda.current_stream()
        side = []
        if n_streams > 1:
            ops._amax_reserve(dev, 64 * (T + 1) * min(per, max(n_local, 1)))  # (no fresh arena inside the forked region)
            if len(self._streams) < n_streams - 1:
                self._streams += [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1 - len(self._streams))]
            side = self._streams[:n_streams - 1]
        lstms = (model.frame_predictor, model.posterior, model.prior)

        for b in range(nb if n_local > 0 else 0):
            s = lo + b * per
            e = lo + (b + 1) * per if b < nb - 1 else hi
            parts = []
            k_parts = n_streams if (e - s) >= 32 * n_streams else 1
            for k in range(k_parts):
                ps, pe = s + (e - s) * k // k_parts, s + (e - s) * (k + 1) // k_parts
                model.init_hidden(batch_size=pe - ps)  # (zero states: made on the main stream, before the fork)
                curr = start_img.expand(pe - ps, -1, -1, -1).contiguous()
                if dontcare_in:
                    curr = ops.ZeroRegion.apply(curr, masks[0, ps - lo:pe - lo].contiguous())
                # (`first`: what the main stream allocated for a part that runs elsewhere stays referenced until the join --
                # the caching allocator hands a freed block back to its OWN stream at once)
                parts.append({"s": ps, "e": pe, "curr": curr, "hidden": [m.hidden for m in lstms],
                              "stream": main if k == 0 else side[k - 1], "first": (curr, [m.hidden for m in lstms])})
            if k_parts > 1:
                fork = torch.cuda.Event()
                fork.record(main)
                for part in parts[1:]:
                    part["stream"].wait_event(fork)
            for t in range(T):
                for part in parts:
                    with torch.cuda.stream(part["stream"]):
                        s_, e_ = part["s"], part["e"]
                        n = e_ - s_
                        ls, le = s_ - lo, e_ - lo  # shard-local rows of states / masks / sum_cost_dev
                        for m, hid in zip(lstms, part["hidden"]):
                            m.hidden = hid
                        curr = part["curr"]
                        ac = actions_dev[s_:e_, t].contiguous()
                        mask = masks[t, ls:le] if cfg.model_use_mask else None
                        state = states[t, ls:le] if cfg.model_use_robot_state else None
                        if cfg.model_use_future_mask:
                            mask = torch.cat([mask, masks[t + 1, ls:le]], 1)
                        if cfg.model_use_future_robot_state:
                            state = (state, states[t + 1, ls:le])
                        shared = t == 0 and shared0
                        x4 = model.forward_maps(curr, mask, state, None, ac, False, sample_mean=cfg.sample_mean,
                                                shared_frame=shared)[0]
                        gi = t if t < len(goal_imgs) else -1
                        add = (not cfg.sparse_cost) or t == T - 1
                        nxt = torch.empty_like(curr)
                        before = sum_cost_dev[ls:le].clone() if ret_step_cost else None
                        # locals: the (possibly copied) operands must outlive the raw-pointer launch
                        next_mask = masks[t + 1, ls:le].contiguous() if (dontcare_in or dontcare_cost) else None
                        goal_mask = goal_masks[gi].contiguous() if (dontcare_cost and goal_masks is not None) else None
                        goal_img = goal_imgs[gi].contiguous()
                        _lib.call(
                            "rac_cem_step_tail", x4.data_ptr(), curr.data_ptr(), _lib.ptr(next_mask) if dontcare_in else None,
                            goal_img.data_ptr(), _lib.ptr(next_mask) if dontcare_cost else None, _lib.ptr(goal_mask),
                            kind, w_world, 1 if (add and w_world != 0) else 0, nxt.data_ptr(),
                            sum_cost_dev[ls:le].data_ptr(), n, H * W, _lib.stream_ptr())
                        if ret_obs:
                            all_obs[s_:e_, t] = nxt.cpu()
                        if ret_step_cost:
                            step_cost[s_:e_, t] = (sum_cost_dev[ls:le] - before).cpu().numpy()
                        part["curr"] = nxt
                        part["hidden"] = [m.hidden for m in lstms]
                        part["keep"] = (x4, curr, next_mask, goal_mask, goal_img, ac, mask, state)
            if k_parts > 1:  # join: the main stream goes on only behind every part (and the parts' tensors live until then)
                for part in parts[1:]:
                    done = torch.cuda.Event()
                    done.record(part["stream"])
                    main.wait_event(done)
            parts = None

        # ---- gather the per-candidate costs: the only collective of a CEM iteration ----
        if getattr(self, "time_gather", False):  # bench.py: time the collective alone (drain the rollouts first)
            torch.cuda.synchronize()
        t_gather = timer.time()
        sum_cost = gather_costs(sum_cost_dev[:n_local], N, world, rank)
        self.last_gather_s = timer.time() - t_gather

        if not suppress_print:
            print("======= Samples Gathered  ======= | >>>> Time taken = %f " % (timer.time() - start_time))
        rollouts = defaultdict(float)
        if opt_traj is not None:
            rollouts["optimal_sum_cost"] = sum_cost[-1]
            if ret_obs:
                rollouts["optimal_obs"] = all_obs[-1].numpy()
            sum_cost = sum_cost[:-1]
        sum_cost = self._refine_elites(sum_cost, action_sequences, start, goal)
        rollouts["sum_cost"] = sum_cost
        if ret_obs:
            topk_idx = np.argsort(sum_cost)[-cfg.topk:]
            rollouts["topk_idx"] = topk_idx
            rollouts["obs"] = all_obs[topk_idx].numpy()
        if ret_step_cost:
            rollouts["step_cost"] = step_cost[:len(sum_cost)]
        return rollouts

"""bench.py -- SVG train frames/sec + CEM candidate-rollouts/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W      (N > 1: starts its own ranks as a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (synthetic 64x64 video; random weights of the named architecture, scaled so that activations stay O(1) and the
predicted frames depend on the actions -- synthetic.synth_state_dict; fp32 storage and accumulation):
  train: BASELINE.json configs[1] -- bs 16 per GPU, n_past 1, n_future 5, g_dim 512, z_dim 64, robot-aware
         flags (mask + future mask + robot state, dontcare_l1).  One step = zero_grad, 5-step BPTT forward,
         losses, backward, (N>1: RCCL gradient all-reduce), fused Adam, loss readback.  `value` = frames/s.
  cem:   configs[2] -- 1000 candidates per GPU x horizon 15 (14 model steps) through the frozen model;
         one "iteration" = generate_model_rollouts (N>1: candidates sharded, one all-gather of the costs).
After the timed regions rank 0 (N = 1) runs the CPU oracle on a bounded sample of the same workloads: that is the
`cpu_baseline`, and the same numbers are asserted against the GPU's (train losses of one step at the full batch,
sum_cost of 64 of the 1000 candidates) -- the benchmark refuses to print a line for a path that has drifted.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _launch_ranks_if_needed():
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is not a rank.  It starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process (never
    an exec), lets the child's rank 0 print the one JSON line on the shared stdout, and exits with the child's return code.
    Runs before torch or the HIP library is imported: nothing in this process ever touches the GPU."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv[1:]):
        if a == "--gpus" and i + 2 < len(sys.argv):
            n = int(sys.argv[i + 2])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: starting {n} ranks as a child torch.distributed.run", file=sys.stderr,
          flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _launch_ranks_if_needed()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from robot_aware_control_amd import ops, synthetic as syn  # noqa: E402
from robot_aware_control_amd.cem import CEMPolicy  # noqa: E402
from robot_aware_control_amd.model import SVGConvModel  # noqa: E402
from robot_aware_control_amd.state import DemoGoalState, State  # noqa: E402
from robot_aware_control_amd.trainer import PredictionTrainer  # noqa: E402

# MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak 2.5 PFLOP/s; the split-precision convs issue 3 fp16 MFMA products
# per fp32 product (two fp16 parts per operand) -> 833.3 TFLOP/s algorithmic; exact-fp32 MFMA 157.3 TFLOP/s
F16_MFMA_PEAK_TFLOPS = 2500.0
SPLIT_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3
F32_MFMA_PEAK_TFLOPS = 157.3
TRAIN_FWD_GFLOP_PER_SAMPLE_STEP = 36.26   # SURVEY.md 8d, hook-counted on the reference (g512/z64, 64x64, RA flags)
CEM_FWD_GFLOP_PER_CAND_STEP = 24.46
CEM_CHECK_TOP, CEM_CHECK_OTHERS = 32, 32  # the oracle re-rolls the GPU's own top 32 candidates and 32 random others



def fwd_gflop(cf, train=True):
    """Algorithmic conv FLOP (2 x MAC, convs and the transposed-conv head only) of ONE forward pass of the model per sample
    and time step: the train forward as the reference runs it (two encoder passes, prior + posterior + frame predictor,
    dynamics.py:575-644) or the planner's prior-only forward.  Reproduces SURVEY.md 8d's hook-counted figures exactly
    (36.26 / 24.46 / 24.45 / 7.42 / 145.03 GF); a NormConvLSTMCell's two g -> 4g convs are one ConvLSTMCell's 2g -> 4g."""
    g, z, A, R = cf.g_dim, cf.z_dim, cf.action_dim, cf.robot_dim
    hw = cf.image_height * cf.image_width
    enc_c = cf.channels + ((1 + (1 if cf.model_use_future_mask else 0)) if cf.model_use_mask else 0)
    if getattr(cf, "model_use_heatmap", False):
        enc_c += 1 + (1 if getattr(cf, "model_use_future_heatmap", False) else 0)
    extra = (R if cf.model_use_robot_state else 0) + (R if cf.model_use_future_robot_state else 0)
    enc = (hw * 9 * (enc_c * 64 + 64 * 64) + hw // 4 * 9 * (64 * 128 + 128 * 128)
           + hw // 16 * 9 * (128 * 256 + 2 * 256 * 256) + hw // 64 * 9 * (256 * 512 + 512 * 512 + 512 * g))
    dec = (hw // 64 * 9 * (g * 512 + 512 * 512 + 512 * 256) + hw // 16 * 9 * (512 * 256 + 256 * 256 + 256 * 128)
           + hw // 4 * 9 * (256 * 128 + 128 * 64) + hw * 9 * (128 * 64 + 64 * (cf.channels + 1)))
    lat = hw // 64
    lstm = lat * (25 + 9) * (2 * g * 4 * g)
    head = lat * 9 * g * z * 2
    prior_in = lat * 9 * (g + A + extra) * g
    post_in = lat * 9 * (g + (R if cf.model_use_robot_state else 0)) * g
    fp_in = lat * 9 * (g + A + z + extra) * g
    if train:
        mac = 2 * enc + dec + 3 * lstm + 2 * head + prior_in + post_in + fp_in
    else:
        mac = enc + dec + 2 * lstm + head + prior_in + fp_in
    return 2.0 * mac / 1e9


RA = dict(model_use_mask=True, model_use_future_mask=True, model_use_robot_state=True,
          reconstruction_loss="dontcare_l1")


def namespace(dev, **kw):
    d = dict(image_width=64, image_height=64, channels=3, g_dim=512, z_dim=64, action_dim=5, robot_dim=5, batch_size=16,
             n_past=1, n_future=5, model_use_future_robot_state=False, model_use_heatmap=False,
             model_use_future_heatmap=False, black_robot_input=False, last_frame_skip=True, robot_pixel_weight=0.0,
             beta=1e-4, lr=1e-4, beta1=0.9, sample_mean=True, lstm_group_norm=False, candidates_batch_size=200,
             sparse_cost=False, reward_type="dense", robot_cost_weight=0.0, world_cost_weight=1.0, topk=5,
             device=dev, debug_cem=False, log_dir="/tmp/rac_bench", img_cost_threshold=None, img_cost_world_norm=True,
             experiment="train_robonet", robot_joint_dim=5, load_movement_info=False, movement_weight=1.0,
             scheduled_sampling=False, scheduled_sampling_k=4000, model="svg", optimizer="adam", seed=0, wandb=False,
             cem_shard=True, ddp_bucket_mb=64, dynamics_model_ckpt=None)
    d.update(RA)
    d.update(kw)
    return argparse.Namespace(**d)


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: the affinity mask, capped at the GPU box's CPU share (16 per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_model():
    """CPU model string of this host (BASELINE.md 4 asks for it beside the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def barrier_sync(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, dev, distributed):
    if not distributed:
        return x
    t = torch.tensor([x], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def per_rank(x, dev, distributed, world):
    """Every rank's value of a scalar (rank 0 reports the list: a slow rank shows up in the SCALE line)."""
    if not distributed:
        return [x]
    t = torch.tensor([x], device=dev, dtype=torch.float64)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def _cdiv(a, b):
    return (a + b - 1) // b


def pmc_traffic(tag, kernel_substr, workgroups):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC profile
    (profiles/*_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        rows = json.load(open(files[-1])).get(tag, [])
    except (OSError, ValueError):
        return None, None
    for r in rows:
        if kernel_substr in r["kernel"] and (workgroups is None or r["workgroups"] == workgroups):
            return r["hbm_side_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def pmc_tail(workload):
    """HBM GB/s of the memory-bound tail kernels (`roofline.tail`): bytes per launch from the newest committed PMC profile,
    durations from the same profile set's kernel trace (profiles/*_pmc_traffic.json, key "tail"; tools/profile_pack.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic.json")))
    if not files:
        return None
    try:
        rows = [r for r in json.load(open(files[-1])).get("tail", []) if r["workload"] == workload]
    except (OSError, ValueError, KeyError):
        return None
    if not rows:
        return None
    out = {r["kernel"]: {"bytes": r["hbm_side_bytes_per_launch"], "avg_ms": r["avg_us"] / 1e3, "gbps": r["gbps"],
                         "frac_of_8TBps": r["frac_of_8TBps"]} for r in rows}
    out["source"] = os.path.relpath(files[-1], ROOT)
    out["note"] = "from the committed profile of this build's kernels (PMC passes + kernel trace), not measured by this run"
    return out


def profile_summary(prof, flops_per_pixel_row):
    """Average HIP-event duration of the profiled kernel and its algorithmic FLOP rate."""
    if not prof["events"]:
        return None
    ms = [e0.elapsed_time(e1) for e0, e1, _ in prof["events"]]
    rows = prof["events"][0][2]
    avg_ms = float(np.mean(ms))
    flop = flops_per_pixel_row * rows
    return {"launches": len(ms), "avg_ms": avg_ms, "tflops": flop / (avg_ms * 1e-3) / 1e12, "flop_per_launch": flop,
            "split": bool(prof.get("split"))}


def kernel_roofline(k):
    peak = SPLIT_PEAK_TFLOPS if k["split"] else F32_MFMA_PEAK_TFLOPS
    return {"arith": ("fp16x2-split operands (22 bits, power-of-two tensor scales), 3 fp16 MFMA products per fp32 "
                      "product, fp32 accumulate" if k["split"] else "exact fp32 MFMA"),
            "peak": peak, "frac": k["tflops"] / peak,
            # what a kernel of nothing but these MFMAs (register-resident operands) sustains under the chip's power cap:
            # 1939 TFLOP/s of fp16 products = 0.776 of nominal (profiles/r03_mfma_power.md); informational, `frac` is the contract's
            **({"sustained_mfma_peak": round(SPLIT_PEAK_TFLOPS * 0.776, 1),
                "frac_of_sustained": k["tflops"] / (SPLIT_PEAK_TFLOPS * 0.776)} if k["split"] else {})}


def phase_breakdown(events, steps):
    """Mean GPU time between the trainer's phase marks over the timed steps (ms)."""
    if not events:
        return None
    per = len(events) // steps
    out = {}
    for s in range(steps):
        chunk = events[s * per:(s + 1) * per]
        for (_, e0), (name, e1) in zip(chunk[:-1], chunk[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) / steps
        if s + 1 < steps:  # readback wait, python between steps, zero_grad of the next step's start mark
            out["between_steps"] = out.get("between_steps", 0.0) + chunk[-1][1].elapsed_time(events[(s + 1) * per][1]) / steps
    return {k: round(v, 3) for k, v in out.items()}


def train_workload_name(args, cf):
    if getattr(args, "deployed", False):
        base = ("SVG train step in the configuration of the reference's deployed checkpoints (evaluate_checkpoint.py:40-46, "
                "roboaware line): 48x64 frames, bs 16/GPU, n_past 1, n_future 5, --lstm_group_norm True, future robot state")
    elif args.cfg5:
        base = "SVG train step, per-GPU shard of BASELINE configs[4]: 128x128, bs 8/GPU, n_past 1, n_future 10"
    elif args.h48:
        base = "SVG train step at the reference's default frame size 48x64 (side config): bs 16/GPU, n_past 1, n_future 5"
    else:
        base = "SVG train step, BASELINE configs[1]: 64x64, bs 16/GPU, n_past 1, n_future 5"
    gn = ", --lstm_group_norm True (NormConvLSTMCell, side config)" if args.group_norm else ""
    sched = getattr(args, "sched", None)
    if sched:
        gn += (", --scheduled_sampling True with the coin forced: " +
               ("every frame after the first fed back from the model's own prediction" if sched == "all"
                else "every other frame fed back (50 % mix)"))
    return f"{base}, g_dim {cf.g_dim}, z_dim {cf.z_dim}, robot-aware flags (dontcare_l1){gn}"


def build_train(args, dev):
    cf = namespace(dev, lstm_group_norm=args.group_norm, ddp_shard_optimizer=bool(getattr(args, "shard_optimizer", False)))
    if getattr(args, "deployed", False):  # evaluate_checkpoint.py:40-46 (roboaware line) at the default 48 x 64 frame
        cf.g_dim, cf.lstm_group_norm, cf.image_height, cf.model_use_future_robot_state = 256, True, 48, True
    if args.h48:  # the reference's default frame size (config/__init__.py:166-171): 48 x 64 -> 6 x 8 latent maps
        cf.image_height = 48
    if args.cfg5:  # BASELINE configs[4] per GPU: 128x128 frames, 8 samples, 10 predicted frames (16x16 latent maps)
        cf.image_width = cf.image_height = 128
        cf.batch_size, cf.n_future = 8, 10
    log(f"building trainer (g{cf.g_dim}/z{cf.z_dim}, {cf.image_height}x{cf.image_width})")
    tr = PredictionTrainer(cf)
    # as PredictionTrainer.train() does: the large weights' optimiser update may finish under the next step's encoder
    # (every step's whole update still lies inside the timed region: it ends with a device-wide synchronise)
    if os.environ.get("RAC_ADAM_OVERLAP", "1") == "1" and hasattr(tr.optimizer, "overlap_next_forward"):
        tr.optimizer.overlap_next_forward = True
    # random weights that keep activations O(1) and make the predictions depend on the actions (synth_state_dict)
    tr.model.load_state_dict(syn.synth_state_dict(tr.model, seed=11))
    tr.model.train()
    return cf, tr


def bench_train(args, dev, rank, world, distributed, exact_of=None):
    """`exact_of`: the result of the split-precision run whose first step (same weights, data, noise) this run repeats
    with every conv on the exact-fp32 MFMA kernels (ops.SPLIT_GEMM off): a shorter timed region, no CPU check."""
    cf, tr = build_train(args, dev)
    B, T = cf.batch_size, cf.n_past + cf.n_future
    side_cfg = args.h48 or args.cfg5 or args.group_norm or getattr(args, "deployed", False) or getattr(args, "sched", None)
    want_check = rank == 0 and world == 1 and not args.no_cpu_baseline and not side_cfg and exact_of is None
    # scheduled sampling with the coin forced (trainer.py:132-147,354): frame i of the window is ground truth or the model's
    # own prediction of it; "all": every frame after the first is fed back (what > 85 % of the README run's steps do)
    n_steps = cf.n_past + cf.n_future - 1
    sched = getattr(args, "sched", None)
    use_truth = None
    if sched == "all":
        use_truth = [True, True] + [False] * (n_steps - 1)
    elif sched == "mix":
        use_truth = [True, True] + [bool(i % 2) for i in range(n_steps - 1)]
    check = None
    if exact_of is not None and exact_of["check"] is not None:
        ck = exact_of["check"]
        tr.model.load_state_dict({k: v.clone() for k, v in ck["sd"].items()})
        sd0, eps = ck["sd"], ck["eps"]
        want_check = True
    elif want_check:  # the very first step runs on recorded noise so that the CPU oracle can repeat it exactly
        sd0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
        eps = syn.synth_eps(seed=77, steps=T - 1, B=B, z=cf.z_dim, h=cf.image_height // 8, w=cf.image_width // 8)
    if want_check:
        queue = [e for pair in eps for e in pair]
        tr.model.eps_source = lambda shape: queue.pop(0)
    batches = [syn.synth_video(seed=100 + rank * 1000 + i, T=T, B=B, H=cf.image_height, W=cf.image_width)
               for i in range(2)]
    batches_dev = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
    for i in range(3):  # prime the caching allocator (new tensor sizes cost a hipMalloc + sync each); not a warmup step
        losses = tr._train_step(batches_dev[i % 2], use_truth=use_truth)
        if i == 0 and want_check:
            check = {"sd": sd0, "data": batches[0], "eps": eps, "gpu_losses": dict(losses)}
            tr.model.eps_source = None
    torch.cuda.synchronize()
    log("allocator primed")
    steps, warmup = (args.steps, args.warmup) if exact_of is None else (args.exact_steps, 2)
    for i in range(warmup):
        tr._train_step(batches_dev[i % 2], use_truth=use_truth)
        torch.cuda.synchronize()
    g = cf.g_dim
    # dominant kernel: ConvLSTM layer-0 gate GEMM, FWD  (M = B*64, N = 4g, K = 25 * 2g)
    prof = {"match": (ops.FWD, 5, 2 * g, 4 * g), "events": []}
    ops.PROFILE = prof
    tr.phase_events = []
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for i in range(steps):
        tr._train_step(batches_dev[i % 2], use_truth=use_truth)
    barrier_sync(distributed)
    dt_local = time.perf_counter() - t0
    opt = tr.optimizer
    opt_info = {"optimizer": type(opt).__name__,
                "shard_requested": bool(getattr(cf, "ddp_shard_optimizer", False)),
                "late_update_overlapped": bool(getattr(opt, "_late_group", lambda: None)() is not None)}
    opt.wait_params()  # nothing of this trainer's optimiser stays installed behind it (ops.PARAM_GATE is process-global)
    if ops.PARAM_GATE is opt:
        ops.PARAM_GATE = None
    dt = max_over_ranks(dt_local, dev, distributed)
    log(f"train{'' if exact_of is None else ' (exact fp32)'}: {steps} steps in {dt:.3f} s")
    ops.PROFILE = None
    events, tr.phase_events = tr.phase_events, None
    starts = [e for n, e in events if n == "start"]
    step_ms = [a.elapsed_time(b) for a, b in zip(starts[:-1], starts[1:])]
    frames = world * B * T * steps
    # SURVEY 8d's hook-counted figures for the BASELINE configs (36.26 / 145.03 GF per sample-step); the analytic count --
    # which reproduces them -- for the side configurations
    if args.cfg5 and side_only(args, "cfg5"):
        per_fwd = 145.03
    elif args.cfg5 or args.h48 or args.group_norm or getattr(args, "deployed", False):
        per_fwd = fwd_gflop(cf)
    else:
        per_fwd = TRAIN_FWD_GFLOP_PER_SAMPLE_STEP
    if not (args.h48 or args.group_norm or getattr(args, "deployed", False)):
        assert abs(fwd_gflop(cf) - per_fwd) < 0.01, (fwd_gflop(cf), per_fwd)
    step_flop = 3 * B * (T - 1) * per_fwd * 1e9
    kern = profile_summary(prof, 2.0 * (4 * g) * (25 * 2 * g))
    return {"frames_per_s": frames / dt, "ms_per_step": dt / steps * 1e3,
            "median_ms_per_step": float(np.median(step_ms)) if step_ms else None,
            "step_tflops_per_gpu": step_flop / (dt / steps) / 1e12, "step_tflop": step_flop / 1e12, "kernel": kern,
            "global_batch": world * B, "phases": phase_breakdown(events, steps), "steps": steps,
            "rank_ms_per_step": [x / steps * 1e3 for x in per_rank(dt_local, dev, distributed, world)],
            "check": check, "cf": cf, "workload": train_workload_name(args, cf), "optimizer": opt_info}


def side_only(args, name):
    """Is `name` the only side switch set (the BASELINE config it names, not a combination)?"""
    flags = {"cfg5": args.cfg5, "h48": args.h48, "group_norm": args.group_norm, "deployed": getattr(args, "deployed", False)}
    return flags[name] and not any(v for k, v in flags.items() if k != name)


class SyntheticRobotInputs:
    """`predict_batch` of the robot-aware planner for the benchmark: AtlasRobotModel (states and masks of every
    candidate from one launch, robot_atlas.py) over the synthetic arm atlas of synthetic.synth_arm_atlas -- the MuJoCo
    renders of the reference's analytical models cannot be produced in this image."""

    def __init__(self, dev):
        self.model = syn.SyntheticArmModel(dev).atlas(108, 121)  # 5 mm grid over the workspace

    def predict_batch(self, data, thick=True):
        return self.model.predict_batch(data, thick)


def bench_cem(args, dev, rank, world, distributed, ra=False, exact_of=None, deployed=False):
    """`ra`: the robot-aware planner (mask + future mask + robot state into the model, dontcare cost; per-candidate
    states / masks produced on the device).  `exact_of`: repeat the split-precision run's problem with the same weights
    on the exact-fp32 MFMA kernels (one timed iteration).  `deployed`: the reference's deployed checkpoints' model
    (evaluate_checkpoint.py:40-46, vanilla line: g 256 / z 64, --lstm_group_norm True) on the default 48 x 64 frame."""
    n_per_gpu, horizon = args.cem_candidates, 15
    flags = dict(RA, reward_type="dontcare") if ra else dict(model_use_mask=False, model_use_future_mask=False,
                                                             model_use_robot_state=False, reconstruction_loss="l1")
    cf = namespace(dev, candidates_batch_size=args.cem_batch, batch_size=args.cem_batch,
                   lstm_group_norm=args.group_norm or deployed, experiment="control_wx250s_synthetic",
                   cem_shared_start=not args.no_cem_shared_start, **flags)
    if deployed:
        cf.g_dim, cf.image_height = 256, 48
    model = SVGConvModel(cf)
    # (action channels amplified 1000x: the eight graded candidates' costs then sit >= 1e-3 apart, relative)
    model.load_state_dict(syn.synth_state_dict(model, seed=12, action_gain=1000.0))
    if exact_of is not None and exact_of["check"] is not None:
        model.load_state_dict({k: v.clone() for k, v in exact_of["check"]["sd"].items()})
    if distributed:
        dist.broadcast(model.flat_parameters()[0], src=0)
    model.eval()
    N = n_per_gpu * world
    demo = None
    if deployed:
        prob = syn.synth_cem_problem(seed=0, N=N, T=horizon - 1, H=cf.image_height, W=cf.image_width)
    elif ra:
        prob = syn.synth_cem_problem(seed=0, N=N, T=horizon - 1)
    else:
        # the planning problem with well-separated elites (K / K+1 cost gap >= 1e-3; the fixture of
        # tests/test_gpu_fullsize.py): eight graded blends towards a demonstration among the N candidates, the per-step
        # goal images = the model's own rollout of that demonstration (rolled out below, outside the timed region)
        prob, demo = syn.demo_problem(False, N, horizon - 1, seed=0)
        prob["actions"] = prob["actions"][:N].contiguous()
    robot = SyntheticRobotInputs(dev) if ra else None
    pol = CEMPolicy(cf, model, horizon=horizon, opt_iter=10, action_candidates=N, topk=5, init_std=0.015,
                    robot_model=robot)
    start = State(img=prob["start_img"], state=np.array([0.28, 0.0, 0.12, 0.0, 0.0], np.float32),
                  qpos=np.zeros(5, np.float32))
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
    if demo is not None:
        if exact_of is not None and exact_of["check"] is not None:  # the split run's goal frames: same problem
            prob["goal_imgs"] = exact_of["check"]["prob"]["goal_imgs"]
        else:
            obs = pol.traj_sampler.generate_model_rollouts(demo[None].clone(), start, goal, ret_obs=True)["obs"][0]
            prob["goal_imgs"] = syn.frames_to_goal_images(obs)
        prob["goal_masks"] = [prob["goal_masks"][0]] * (horizon - 1)
        goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
    g = cf.g_dim
    tag = ("cem-deployed" if deployed else "cem-ra" if ra else "cem") + ("" if exact_of is None else " (exact fp32)")
    log(f"{tag}: model built, {N} candidates")
    small = syn.synth_cem_problem(seed=0, N=N, T=2, H=cf.image_height, W=cf.image_width)
    pol.traj_sampler.generate_model_rollouts(small["actions"], start, goal)  # allocator priming, 2 model steps
    iters, warm = (args.cem_iters, args.cem_warmup) if exact_of is None else (1, 0)
    for _ in range(warm):
        pol.traj_sampler.generate_model_rollouts(prob["actions"], start, goal)
        log(f"{tag} warmup iteration done")
    prof = {"match": (ops.FWD, 5, 2 * g, 4 * g), "events": []}
    ops.PROFILE = prof
    pol.traj_sampler.time_gather = distributed
    gather_s = []
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for _ in range(iters):
        ro = pol.traj_sampler.generate_model_rollouts(prob["actions"], start, goal)
        gather_s.append(pol.traj_sampler.last_gather_s)
    barrier_sync(distributed)
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, dev, distributed)
    log(f"{tag}: {iters} iterations in {dt:.3f} s")
    ops.PROFILE = None
    pol.traj_sampler.time_gather = False
    assert len(ro["sum_cost"]) == N and np.all(np.isfinite(ro["sum_cost"]))
    per_fwd = fwd_gflop(cf, train=False) if (deployed or args.group_norm) else CEM_FWD_GFLOP_PER_CAND_STEP
    it_flop_per_gpu = n_per_gpu * (horizon - 1) * per_fwd * 1e9
    kern = profile_summary(prof, 2.0 * (4 * g) * (25 * 2 * g))
    out = {"rollouts_per_s": N * iters / dt, "s_per_iter": dt / iters, "iters": iters,
           "tflops_per_gpu": it_flop_per_gpu / (dt / iters) / 1e12, "kernel": kern, "candidates": N,
           "candidates_batch_size": args.cem_batch, "check": None,
           "rank_s_per_iter": [x / iters for x in per_rank(dt_local, dev, distributed, world)],
           "cost_allgather_ms": float(np.mean(gather_s)) * 1e3 if distributed else None, "get_action": None}
    out["gflop_per_candidate_step"] = per_fwd
    if exact_of is not None:
        if exact_of["check"] is not None:
            out["check_sum_cost"] = ro["sum_cost"][exact_of["check"]["idx"]].copy()
        return out
    if deployed:
        return out
    # secondary metric: the whole planner call at the config's opt_iter -- sampling, robot inputs, rollouts,
    # cost gather, top-k, refit
    pol.optimization_iter = args.cem_opt_iter
    barrier_sync(distributed)
    t1 = time.perf_counter()
    pol.get_action(start, goal, 0, 0)
    barrier_sync(distributed)
    dt_ga = max_over_ranks(time.perf_counter() - t1, dev, distributed)
    out["get_action"] = {"value": N * args.cem_opt_iter / dt_ga, "unit": "candidate-rollouts/s",
                         "opt_iter": args.cem_opt_iter, "s_per_call": dt_ga,
                         "note": "CEMPolicy.get_action end to end (sampling, rollouts, cost gather, top-k, refit)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and N >= 1000 and not args.group_norm:
        # the oracle re-rolls the GPU's OWN best candidates (the elite set must be right, not 64 arbitrary costs) + others
        order = np.argsort(-ro["sum_cost"], kind="stable")
        idx = np.concatenate([order[:CEM_CHECK_TOP],
                              np.random.RandomState(0).choice(order[CEM_CHECK_TOP:], CEM_CHECK_OTHERS, replace=False)])
        out["check"] = {"sd": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "prob": prob,
                        "idx": idx, "gpu_sum_cost": ro["sum_cost"][idx].copy(), "gpu_top": order[:5].copy(), "ra": ra}
        if ra:
            # the robot model's answer for the checked candidates (a candidate's states and masks depend on nothing but its
            # own actions): what the oracle's rollout of them is fed
            st, mk = pol.traj_sampler._predict_robot(prob["actions"][torch.from_numpy(idx)], start, len(idx), horizon - 1)
            out["check"]["states"], out["check"]["masks"] = st.float().cpu(), mk.float().cpu()
    return out


def cpu_baseline(train, cem):
    """The oracle (CPU restatement of the reference path, pinned by tests/golden) on this host's cores: a bounded sample
    of the same workloads -- configs[1] at its full batch (1 warm-up step + 3 timed), 64 of the 1000 candidates of
    configs[2] in one pass (the GPU's own top 32 and 32 random others) -- and, with the same numbers, the parity check
    of what was just benchmarked: losses <= 1e-4, costs <= 1e-5, the oracle's top 5 equal to the GPU's."""
    from oracle import svg_oracle as orc
    torch.set_num_threads(host_threads())
    cores = torch.get_num_threads()
    out = {"unit": "frames/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port", "checked": {}}
    sample = []
    if train is not None and train["check"] is not None:
        ck, cf = train["check"], train["cf"]
        B, T = cf.batch_size, cf.n_past + cf.n_future
        cfg = orc.Cfg(g_dim=cf.g_dim, z_dim=cf.z_dim, batch_size=B, n_past=cf.n_past, n_future=cf.n_future, lr=cf.lr, **RA)
        ts = orc.TrainState.create(cfg, ck["sd"])
        log(f"cpu baseline on {cores} threads: train step 0 (warm-up; the GPU's first step repeated on its noise)")
        ref = orc.train_step(ts, ck["data"], ck["eps"], None, do_update=True)
        worst = max(abs(ck["gpu_losses"][k] - ref[k]) / (abs(ref[k]) + 1e-12) for k in ref)
        assert worst <= 1e-4, ("GPU train step drifted from the oracle", ck["gpu_losses"], ref)
        out["checked"]["train_losses_rel_err"] = worst
        ck["oracle_losses"] = dict(ref)
        times = []
        for i in (1, 2, 3):
            data = syn.synth_video(seed=100 + i % 2, T=T, B=B)
            t0 = time.perf_counter()
            orc.train_step(ts, data, None, None, do_update=True)
            times.append(time.perf_counter() - t0)
            log(f"cpu baseline train step {i}: {times[-1]:.1f} s")
        out["value"] = B * T / float(np.mean(times))
        sample.append(f"train: configs[1] at batch {B}, 1 warm-up + {len(times)} timed optimiser steps "
                      f"({np.mean(times):.1f} s each, min {min(times):.1f}, max {max(times):.1f})")
    if cem is not None and cem["check"] is not None:
        ck = cem["check"]
        idx = ck["idx"]
        n = len(idx)
        ccfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=n, candidates_batch_size=n, sample_mean=True, reward_type="dense",
                       model_use_mask=False, model_use_future_mask=False, model_use_robot_state=False,
                       reconstruction_loss="l1")
        prob = ck["prob"]
        # warm the CPU leg (thread pool, conv primitive caches) on two candidates x two steps before the timed pass
        wcfg = orc.Cfg(**{**ccfg.__dict__, "batch_size": 2, "candidates_batch_size": 2})
        orc.cem_rollouts(ck["sd"], wcfg, prob["actions"][idx[:2], :2], prob["start_img"], prob["goal_imgs"][:2],
                         prob["goal_masks"][:2])
        t0 = time.perf_counter()
        ref = orc.cem_rollouts(ck["sd"], ccfg, prob["actions"][idx], prob["start_img"], prob["goal_imgs"],
                               prob["goal_masks"])["sum_cost"]
        t_cem = time.perf_counter() - t0
        ck["oracle_sum_cost"] = ref
        err = float(np.abs(ck["gpu_sum_cost"] - ref).max() / np.abs(ref).max())
        assert err <= 1e-5, ("GPU rollout costs drifted from the oracle", err)
        out["checked"]["cem_sum_cost_rel_err"] = err
        # elite set (north_star: "CEM elite indices are bit-exact"): on this fixture the GPU's ranks 1..6 are >= 1e-3 apart
        # (asserted), so the oracle's ranking of the GPU's top 32 must start with the GPU's top 5, indices AND order --
        # unconditionally
        top_ref = idx[np.argsort(-ref, kind="stable")][:5]
        g = np.sort(ck["gpu_sum_cost"])[::-1]
        gaps = (g[:5] - g[1:6]) / np.abs(g[1:6])
        out["checked"]["cem_top5_identical"] = bool(all(int(a) == int(b) for a, b in zip(top_ref, ck["gpu_top"])))
        out["checked"]["cem_top5_min_gap_rel"] = float(gaps.min())
        out["checked"]["cem_rank5_rank6_gap_rel"] = float(gaps[4])
        out["checked"]["cem_fixture"] = "synthetic.demo_problem: graded blends towards a demonstration whose rollout is the goal"
        assert gaps.min() >= 1e-3, ("the benchmark's planning fixture no longer separates its elites", gaps)
        assert out["checked"]["cem_top5_identical"], ("GPU elite set differs from the oracle's", top_ref, ck["gpu_top"], gaps, err)
        out["cem_value"], out["cem_unit"] = n / t_cem, "candidate-rollouts/s"
        sample.append(f"cem: {n} of the 1000 candidates (the GPU's top {CEM_CHECK_TOP} + {CEM_CHECK_OTHERS} others) x 14 "
                      f"steps in one pass ({t_cem:.1f} s, after a 2-candidate x 2-step warm-up pass)")
        if "value" not in out:
            out["value"], out["unit"] = out["cem_value"], out["cem_unit"]
    out["sample"] = "; ".join(sample)
    return out


def check_cem_ra(cem_ra):
    """The robot-aware planner's costs against the oracle: the GPU's own top 32 + 32 others re-rolled on the CPU with the
    states and masks the device-side robot model gave those candidates; sum_cost <= 1e-5 of the largest, and the ranking of
    the GPU's elites wherever neighbours are further apart than 10 x the error."""
    from oracle import svg_oracle as orc
    ck = cem_ra["check"]
    idx = ck["idx"]
    n = len(idx)
    ccfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=n, candidates_batch_size=n, sample_mean=True, reward_type="dontcare", **RA)
    prob = ck["prob"]
    t0 = time.perf_counter()
    ref = orc.cem_rollouts(ck["sd"], ccfg, prob["actions"][idx], prob["start_img"], prob["goal_imgs"], prob["goal_masks"],
                           ck["states"], ck["masks"])["sum_cost"]
    t_cem = time.perf_counter() - t0
    got = ck["gpu_sum_cost"]
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert err <= 1e-5, ("robot-aware GPU rollout costs drifted from the oracle", err)
    order = np.argsort(-ref[:CEM_CHECK_TOP], kind="stable")
    gaps = np.abs(np.diff(ref[:CEM_CHECK_TOP][order])) / np.abs(ref).max()
    resolvable = gaps[:5] > 10 * err
    same = all(got[order[i]] > got[order[i + 1]] for i in np.nonzero(resolvable)[0])
    assert same, ("robot-aware GPU elite order differs from the oracle's where the gap resolves it", gaps[:5], err)
    return {"cem_ra_sum_cost_rel_err": err, "cem_ra_top5_gaps_resolved": int(resolvable.sum()),
            "cem_ra_top5_order_identical_where_resolved": bool(same), "cem_ra_top5_min_gap_rel": float(gaps[:5].min()),
            "cem_ra_cpu_rollouts_per_s": n / t_cem}


def exact_fp32_runs(args, dev, rank, world, distributed, train, cem):
    """The same workloads with every conv on the exact-fp32 MFMA kernels (ops.SPLIT_GEMM off; v_mfma_f32_32x32x2_f32,
    peak 157.3 TFLOP/s): what the split-precision operand format buys, and what it costs in accuracy -- the first train
    step and the checked rollout costs of both builds side by side (and against the oracle when it ran)."""
    keep = ops.SPLIT_GEMM
    ops.SPLIT_GEMM = False
    try:
        out, err = {"arith": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), same kernels otherwise", "peak": F32_MFMA_PEAK_TFLOPS}, {}
        if train is not None:
            ex = bench_train(args, dev, rank, world, distributed, exact_of=train)
            torch.cuda.empty_cache()
            k = ex["kernel"] or {"tflops": None, "avg_ms": None}
            out["train"] = {"ms_per_step": ex["ms_per_step"], "frames_per_s": ex["frames_per_s"], "steps": ex["steps"],
                            "step_tflops": ex["step_tflops_per_gpu"], "step_frac": ex["step_tflops_per_gpu"] / F32_MFMA_PEAK_TFLOPS,
                            "gate_gemm_tflops": k["tflops"], "gate_gemm_avg_launch_ms": k["avg_ms"],
                            "gate_gemm_frac": (k["tflops"] / F32_MFMA_PEAK_TFLOPS) if k["tflops"] else None}
            if train["check"] is not None and ex["check"] is not None:
                a, b = train["check"]["gpu_losses"], ex["check"]["gpu_losses"]
                err["train_losses_split_vs_exact_fp32"] = max(abs(a[q] - b[q]) / (abs(b[q]) + 1e-12) for q in b)
                train["check"]["exact_losses"] = dict(b)
        if cem is not None:
            ex = bench_cem(args, dev, rank, world, distributed, exact_of=cem)
            torch.cuda.empty_cache()
            k = ex["kernel"] or {"tflops": None, "avg_ms": None}
            out["cem"] = {"rollouts_per_s": ex["rollouts_per_s"], "s_per_iteration": ex["s_per_iter"], "iterations": ex["iters"],
                          "tflops": ex["tflops_per_gpu"], "frac": ex["tflops_per_gpu"] / F32_MFMA_PEAK_TFLOPS,
                          "gate_gemm_tflops": k["tflops"], "gate_gemm_avg_launch_ms": k["avg_ms"],
                          "gate_gemm_frac": (k["tflops"] / F32_MFMA_PEAK_TFLOPS) if k["tflops"] else None}
            if cem["check"] is not None and "check_sum_cost" in ex:
                a, b = cem["check"]["gpu_sum_cost"], ex["check_sum_cost"]
                err["cem_sum_cost_split_vs_exact_fp32"] = float(np.abs(a - b).max() / np.abs(b).max())
                cem["check"]["exact_sum_cost"] = b
        return out, err
    finally:
        ops.SPLIT_GEMM = keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="both", choices=["both", "train", "cem"])
    ap.add_argument("--cem-candidates", type=int, default=1000, help="candidates per GPU")
    ap.add_argument("--cem-batch", type=int, default=1000, help="candidates per GPU pass")
    ap.add_argument("--cem-iters", type=int, default=2)
    ap.add_argument("--cem-warmup", type=int, default=1)
    ap.add_argument("--cem-opt-iter", type=int, default=10, help="CEM iterations of the timed get_action call")
    ap.add_argument("--exact-steps", type=int, default=5, help="timed train steps of the exact-fp32 comparison run")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32 comparison runs")
    ap.add_argument("--no-cem-ra", action="store_true", help="skip the robot-aware planner workload")
    ap.add_argument("--no-cem-shared-start", action="store_true",
                    help="planner step 0: run the encoder on every candidate's copy of the start frame (as the reference does)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h48", action="store_true",
                    help="train workload on 48x64 frames, the reference's default --image_height (not the headline)")
    ap.add_argument("--cfg5", action="store_true",
                    help="train workload at BASELINE configs[4] per-GPU size (128x128, bs 8, n_future 10); not the headline")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: reduce-scatter + Adam on 1/N slices + parameter all-gather (optim.ShardedAdam) instead of "
                         "all-reduce + the fused Adam pass on every rank (not the default: unmeasured on multi-GPU hardware)")
    ap.add_argument("--no-ddp-modes", action="store_true",
                    help="N > 1: time only the chosen gradient-exchange mode (default: both, reported under `ddp_modes`)")
    ap.add_argument("--no-side", action="store_true", help="skip the configs[4] per-GPU side line (`side.cfg5`)")
    ap.add_argument("--side-steps", type=int, default=5, help="timed steps of the configs[4] per-GPU side line")
    ap.add_argument("--deployed", action="store_true",
                    help="train workload in the deployed checkpoints' configuration (g 256, --lstm_group_norm True, 48x64, "
                         "future robot state); not the headline")
    ap.add_argument("--sched", default=None, choices=["all", "mix"],
                    help="train workload with the scheduled-sampling coin forced: every frame after the first fed back "
                         "(all) or every other one (mix); not the headline")
    ap.add_argument("--group-norm", action="store_true",
                    help="both workloads with --lstm_group_norm True (NormConvLSTMCell; not the headline config)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # (RAC_DIST_FORCE=1: a one-rank process group -- every collective of the path runs through RCCL on a one-GPU box)
    distributed = world > 1 or os.environ.get("RAC_DIST_FORCE", "0") == "1"
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment (run `python bench.py "
                         f"--gpus {args.gpus}` with no WORLD_SIZE set and it starts its own ranks, or launch it with "
                         f"torch.distributed.run --nproc-per-node {args.gpus})")
    backend = os.environ.get("RAC_DIST_BACKEND", "nccl")  # "gloo" + RAC_BENCH_ONE_GPU=1: rehearse N ranks on one GPU
    if os.environ.get("RAC_BENCH_ONE_GPU") == "1":
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    elif torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} device(s): one rank "
                         f"per GPU (RAC_BENCH_ONE_GPU=1 + RAC_DIST_BACKEND=gloo rehearses N ranks on one card)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(1234)
    np.random.seed(1234)
    os.environ.setdefault("RAC_GC_FREEZE", "1")  # the benchmark process builds its objects once
    dist_info = None
    if distributed:
        # what the collective library itself saw: its backend, its world size, and the card behind every rank
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": dist.get_rank(), "index": dev.index, "uuid": str(getattr(props, "uuid", "")), "name": props.name}
        cards = [None] * dist.get_world_size()
        dist.all_gather_object(cards, mine)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "devices": sorted(cards, key=lambda c: c["rank"])}
        if os.environ.get("RAC_BENCH_ONE_GPU") != "1" and len({(c["index"], c["uuid"]) for c in cards}) != len(cards):
            raise SystemExit(f"bench.py: two ranks share a device: {cards}")

    out = {"metric": "SVG train frames/sec + CEM candidate-rollouts/sec, 64x64", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "data": "synthetic",
           "dtype": "fp32 (conv operands as two fp16 parts = 22 significant bits, fp32 accumulate; see fp32_exact)"}
    train = cem = None
    ddp_modes = None
    if args.workload in ("both", "train"):
        train = bench_train(args, dev, rank, world, distributed)
        torch.cuda.empty_cache()
        if distributed and not args.no_ddp_modes:
            # both gradient-exchange designs on the same ranks, same workload: all-reduce + the full fused Adam on every
            # rank, and reduce-scatter + Adam on 1/N slices + parameter all-gather.  The headline is the mode
            # the flags chose (all-reduce unless --shard-optimizer); a failure of the other mode is recorded, not fatal.
            def mode_line(t):
                ph = t["phases"] or {}
                line = {"ms_per_step": t["ms_per_step"], "frames_per_s": t["frames_per_s"],
                        "allreduce_exposed": ph.get("allreduce_exposed"), "adam": ph.get("adam"),
                        "rank_ms_per_step": t["rank_ms_per_step"], "optimizer": t["optimizer"]["optimizer"]}
                if t["optimizer"]["shard_requested"] and t["optimizer"]["optimizer"] != "ShardedAdam":
                    line["fallback"] = "sharded optimiser not available at this world size: this line IS all-reduce + full Adam"
                return line
            this, other = ("sharded", "allreduce") if args.shard_optimizer else ("allreduce", "sharded")
            ddp_modes = {this: mode_line(train), "headline": this}
            oa = argparse.Namespace(**vars(args))
            oa.shard_optimizer = not args.shard_optimizer
            # the other mode may fail on ONE rank (out of memory, a collective error): the ranks agree on the outcome before
            # anyone records it -- a rank that failed alone would otherwise leave the rest inside a collective
            res, failure = None, None
            try:
                res = bench_train(oa, dev, rank, world, distributed)
            except Exception as e:  # noqa: BLE001
                failure = f"{type(e).__name__}: {e}"
            flag = torch.tensor([0 if failure is None else 1], device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            n_failed = int(flag.item())
            if n_failed == 0:
                ddp_modes[other] = mode_line(res)
            elif n_failed == dist.get_world_size():
                ddp_modes[other] = {"error": failure}
            else:  # the ranks disagree: whatever state the group is in, this run is not a measurement
                print(f"[bench] rank {rank}: the second gradient-exchange mode failed on {n_failed} of "
                      f"{dist.get_world_size()} ranks ({failure})", file=sys.stderr, flush=True)
                os._exit(3)
            torch.cuda.empty_cache()
    cem_ra = None
    if args.workload in ("both", "cem"):
        cem = bench_cem(args, dev, rank, world, distributed, deployed=args.deployed)
        torch.cuda.empty_cache()
        if not args.no_cem_ra and not args.group_norm and not args.deployed:
            cem_ra = bench_cem(args, dev, rank, world, distributed, ra=True)
            torch.cuda.empty_cache()
    side = None
    headline = not (args.h48 or args.cfg5 or args.group_norm or args.deployed or args.sched)
    if train is not None and not args.no_side and headline:
        # BASELINE configs[4] per GPU (128x128 frames, bs 8, n_future 10: the shard one of 8 DDP ranks trains on), driver-timed
        sa = argparse.Namespace(**vars(args))
        sa.cfg5, sa.steps, sa.warmup = True, args.side_steps, 2
        st = bench_train(sa, dev, rank, world, distributed)
        torch.cuda.empty_cache()
        side = {"cfg5": {"value": st["frames_per_s"], "unit": "frames/s", "ms_per_step": st["ms_per_step"],
                         "steps": st["steps"], "warmup": 2,
                         "config": {"workload": st["workload"], "global_batch": st["global_batch"],
                                    "parallelism": f"ddp{world}",
                                    "algorithmic_tflop_per_step_per_gpu": st["step_tflop"]},
                         "step_achieved": st["step_tflops_per_gpu"], "step_peak": SPLIT_PEAK_TFLOPS,
                         "step_frac": st["step_tflops_per_gpu"] / SPLIT_PEAK_TFLOPS,
                         "time_breakdown_ms": st["phases"]}}
    if train is not None and not args.no_side and headline:
        # the paths the reference's own command lines take (README.md:103,111: --scheduled_sampling True;
        # evaluate_checkpoint.py:40-46: --lstm_group_norm True, g 256; config/__init__.py:165-171: 48 x 64 frames)
        def train_side(**switches):
            sa = argparse.Namespace(**vars(args))
            sa.steps, sa.warmup = args.side_steps, 2
            for k_, v_ in switches.items():
                setattr(sa, k_, v_)
            st = bench_train(sa, dev, rank, world, distributed)
            torch.cuda.empty_cache()
            return {"value": st["frames_per_s"], "unit": "frames/s", "ms_per_step": st["ms_per_step"],
                    "steps": st["steps"], "warmup": 2,
                    "config": {"workload": st["workload"], "global_batch": st["global_batch"],
                               "parallelism": f"ddp{world}", "algorithmic_tflop_per_step_per_gpu": st["step_tflop"]},
                    "step_achieved": st["step_tflops_per_gpu"], "step_peak": SPLIT_PEAK_TFLOPS,
                    "step_frac": st["step_tflops_per_gpu"] / SPLIT_PEAK_TFLOPS, "time_breakdown_ms": st["phases"]}
        side = side or {}
        side["sched"] = train_side(sched="all")
        side["sched"]["vs_teacher_forced"] = side["sched"]["ms_per_step"] / train["ms_per_step"]
        side["sched_mix"] = train_side(sched="mix")
        side["sched_mix"]["vs_teacher_forced"] = side["sched_mix"]["ms_per_step"] / train["ms_per_step"]
        side["deployed_train"] = train_side(deployed=True)
        side["deployed_train_sched"] = train_side(deployed=True, sched="all")
    if cem is not None and not args.no_side and headline:
        dc = bench_cem(args, dev, rank, world, distributed, deployed=True)
        torch.cuda.empty_cache()
        side = side or {}
        side["deployed_cem"] = {"value": dc["rollouts_per_s"], "unit": "candidate-rollouts/s",
                                "s_per_iteration": dc["s_per_iter"],
                                "config": {"workload": "CEM rollouts through the reference's deployed model (evaluate_checkpoint.py:"
                                                       "40-46, vanilla line): g 256 / z 64, --lstm_group_norm True, 48x64 frames, "
                                                       "1000 candidates/GPU x horizon 15, dense image cost",
                                           "candidates": dc["candidates"], "candidates_batch_size": dc["candidates_batch_size"],
                                           "parallelism": f"candidate-shard{world}",
                                           "algorithmic_gflop_per_candidate_step": dc["gflop_per_candidate_step"]},
                                "achieved_tflops_per_gpu": dc["tflops_per_gpu"],
                                "frac_of_split_peak": dc["tflops_per_gpu"] / SPLIT_PEAK_TFLOPS}
    exact = arith_err = None
    if not args.no_exact and headline:
        exact, arith_err = exact_fp32_runs(args, dev, rank, world, distributed, train, cem)

    if train is not None:
        out.update(value=train["frames_per_s"], unit="frames/s", ms_per_step=train["ms_per_step"],
                   median_ms_per_step=train["median_ms_per_step"],
                   config={"workload": train["workload"],
                           "global_batch": train["global_batch"],
                           "parallelism": f"ddp{world}" + ("-sharded-optimizer" if args.shard_optimizer and distributed else ""),
                           "algorithmic_tflop_per_step_per_gpu": train["step_tflop"]})
        k = train["kernel"]
        if k is None:  # no launch of the profiled shape (e.g. --group-norm: separate ih / hh gate convs)
            k = {"split": True, "tflops": train["step_tflops_per_gpu"], "avg_ms": None, "launches": 0}
        kr = kernel_roofline(k)
        traffic, traffic_src = pmc_traffic("gemm_train", "conv16_tile_kernel", None)
        # step level: algorithmic FLOP of the step (the reference-faithful count, second encoder pass included) over wall
        # time, against the peak of the pipe that carries the convs (profiles/*_shapes.md: > 97 % of the conv FLOP runs
        # split-precision; the exact-fp32 remainder is the 5-channel first layer's and the output head's gradients)
        out["roofline"] = {"bound": "mfma", "kernel": "conv16_tile_kernel: FWD 5x5 ConvLSTM gate conv as GEMM "
                           "(M=1024, N=2048, K=25600)", "arith": kr["arith"],
                           "achieved": k["tflops"], "peak": kr["peak"], "unit": "TFLOP/s", "frac": kr["frac"],
                           "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_ms": k["avg_ms"], "launches": k["launches"],
                           "step_achieved": train["step_tflops_per_gpu"], "step_peak": SPLIT_PEAK_TFLOPS,
                           "step_frac": train["step_tflops_per_gpu"] / SPLIT_PEAK_TFLOPS}
        for key in ("sustained_mfma_peak", "frac_of_sustained"):  # informational (profiles/r03_mfma_power.md)
            if key in kr:
                out["roofline"][key] = kr[key]
        if headline:
            out["roofline"]["kernel"] = ("conv16_tile_kernel: FWD 5x5 ConvLSTM gate conv as GEMM (M=%d, N=2048, K=25600)"
                                         % (train["cf"].batch_size * 64))
        tail = pmc_tail("train")
        if tail:
            out["roofline"]["tail"] = tail
        out["time_breakdown_ms"] = train["phases"]
        out["optimizer"] = train["optimizer"]
        if train["optimizer"]["late_update_overlapped"]:
            # (optim.FusedAdam.overlap_next_forward: `adam` is the part on the main stream; the large weights' update runs on
            # a side stream under the next step's `forward`, which it stretches)
            out["time_breakdown_note"] = "adam: main-stream part only; the large weights' update overlaps the next forward"
        if distributed:
            out["ranks"] = {"train_ms_per_step": train["rank_ms_per_step"]}
            out["dist"] = dist_info
        if ddp_modes is not None:
            out["ddp_modes"] = ddp_modes
        # the SCALE record's per-N fields: the driver computes efficiency from the per-N values itself
        out["per_gpu_value"] = train["frames_per_s"] / world
        out["efficiency_vs_n1"] = None
    if cem is not None:
        k = cem["kernel"]
        if k is None:
            k = {"split": True, "tflops": cem["tflops_per_gpu"], "avg_ms": None, "launches": 0}
        gate = {"avg_launch_ms": k["avg_ms"], "tflops": k["tflops"], "launches": k["launches"]}
        gate.update(kernel_roofline(k))
        gate["traffic"], gate["traffic_source"] = pmc_traffic("gemm_cem", "conv16_tile_kernel", None)
        cem_obj = {"value": cem["rollouts_per_s"], "unit": "candidate-rollouts/s", "s_per_iteration": cem["s_per_iter"],
                   "config": {"workload": "CEM rollouts, BASELINE configs[2]: 1000 candidates/GPU x horizon 15 "
                                          "(14 model steps), frozen g512/z64 model, 64x64, dense image cost",
                              "candidates": cem["candidates"], "candidates_batch_size": cem["candidates_batch_size"],
                              "parallelism": f"candidate-shard{world}",
                              # step 0 of a rollout sees the same start frame for every candidate: its encoder pass runs
                              # once (same bits; 1.4 % of the algorithmic FLOPs, which stay in the numerator as the
                              # shared second encoder pass of training does).  --no-cem-shared-start: per candidate
                              "shared_start_frame": not args.no_cem_shared_start},
                   "achieved_tflops_per_gpu": cem["tflops_per_gpu"],
                   "frac_of_split_peak": cem["tflops_per_gpu"] / SPLIT_PEAK_TFLOPS,
                   "gate_gemm": gate, "get_action": cem["get_action"],
                   "per_gpu_value": cem["rollouts_per_s"] / world, "efficiency_vs_n1": None}
        tail = pmc_tail("cem")
        if tail:
            cem_obj["tail"] = tail
        if distributed:
            cem_obj["ranks"] = {"s_per_iteration": cem["rank_s_per_iter"], "cost_allgather_ms": cem["cost_allgather_ms"]}
        if train is None:
            out.update(value=cem["rollouts_per_s"], unit="candidate-rollouts/s", ms_per_step=cem["s_per_iter"] * 1e3,
                       config=cem_obj["config"],
                       roofline={"bound": "mfma", "kernel": "conv16_tile_kernel: FWD 5x5 ConvLSTM gate conv as GEMM "
                                 "(M=64000, N=2048, K=25600)", "arith": gate["arith"], "achieved": k["tflops"],
                                 "peak": gate["peak"], "unit": "TFLOP/s", "frac": gate["frac"],
                                 "traffic": gate["traffic"], "traffic_source": gate["traffic_source"]})
        out["cem"] = cem_obj
    if cem_ra is not None:
        k = cem_ra["kernel"] or {"split": True, "tflops": cem_ra["tflops_per_gpu"], "avg_ms": None, "launches": 0}
        out["cem_ra"] = {"value": cem_ra["rollouts_per_s"], "unit": "candidate-rollouts/s",
                         "s_per_iteration": cem_ra["s_per_iter"],
                         "config": {"workload": "robot-aware CEM rollouts: configs[2] geometry with mask + future mask + robot "
                                                "state into the model, dontcare cost; states and masks of all candidates from "
                                                "AtlasRobotModel on the device (synthetic arm atlas, 5 mm grid), inside the "
                                                "timed region",
                                    "candidates": cem_ra["candidates"], "candidates_batch_size": cem_ra["candidates_batch_size"],
                                    "parallelism": f"candidate-shard{world}"},
                         "achieved_tflops_per_gpu": cem_ra["tflops_per_gpu"],
                         "frac_of_split_peak": cem_ra["tflops_per_gpu"] / SPLIT_PEAK_TFLOPS,
                         "gate_gemm_tflops": k["tflops"], "get_action": cem_ra["get_action"]}
        if distributed:
            out["cem_ra"]["ranks"] = {"s_per_iteration": cem_ra["rank_s_per_iter"],
                                      "cost_allgather_ms": cem_ra["cost_allgather_ms"]}
    if side is not None:
        out["side"] = side
    if exact is not None:
        out["fp32_exact"] = exact
    if distributed and "dist" not in out:
        out["dist"] = dist_info
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(train, cem)
        if cem_ra is not None and cem_ra["check"] is not None:
            cb.setdefault("checked", {}).update(check_cem_ra(cem_ra))  # (the robot-aware line's own check)
        if cb.get("sample"):
            out["cpu_baseline"] = cb
        # split-precision vs exact fp32 vs the oracle, on the same first train step and the same checked candidates
        if arith_err is not None:
            if train is not None and train["check"] is not None and "oracle_losses" in train["check"]:
                ck = train["check"]
                rel = lambda a, b: max(abs(a[q] - b[q]) / (abs(b[q]) + 1e-12) for q in b)
                arith_err["train_losses_split_vs_oracle"] = rel(ck["gpu_losses"], ck["oracle_losses"])
                if "exact_losses" in ck:
                    arith_err["train_losses_exact_fp32_vs_oracle"] = rel(ck["exact_losses"], ck["oracle_losses"])
            if cem is not None and cem["check"] is not None and "oracle_sum_cost" in cem["check"]:
                ck = cem["check"]
                rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
                arith_err["cem_sum_cost_split_vs_oracle"] = rel(ck["gpu_sum_cost"], ck["oracle_sum_cost"])
                if "exact_sum_cost" in ck:
                    arith_err["cem_sum_cost_exact_fp32_vs_oracle"] = rel(ck["exact_sum_cost"], ck["oracle_sum_cost"])
    if arith_err:
        arith_err["note"] = ("max relative error: train = the first optimiser step's loss terms, cem = sum_cost of the 64 "
                             "checked candidates (relative to the largest)")
        out["arith_error"] = arith_err
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

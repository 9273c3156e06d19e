"""bench.py -- SVG train frames/sec + CEM candidate-rollouts/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (synthetic 64x64 video, random-init weights of the named architecture, fp32):
  train: BASELINE.json configs[1] -- bs 16 per GPU, n_past 1, n_future 5, g_dim 512, z_dim 64, robot-aware
         flags (mask + future mask + robot state, dontcare_l1).  One step = zero_grad, 5-step BPTT forward,
         losses, backward, (N>1: RCCL gradient all-reduce), fused Adam, loss readback.  `value` = frames/s.
  cem:   configs[2] -- 1000 candidates per GPU x horizon 15 (14 model steps) through the frozen model;
         one "iteration" = generate_model_rollouts (N>1: candidates sharded, one all-gather of the costs).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from robot_aware_control_amd import ops, synthetic as syn  # noqa: E402
from robot_aware_control_amd.cem import CEMPolicy  # noqa: E402
from robot_aware_control_amd.model import SVGConvModel  # noqa: E402
from robot_aware_control_amd.state import DemoGoalState, State  # noqa: E402
from robot_aware_control_amd.trainer import PredictionTrainer  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak; the split path issues 6 bf16 MFMA products per fp32 product
TRAIN_FWD_GFLOP_PER_SAMPLE_STEP = 36.26   # SURVEY.md 8d, hook-counted on the reference (g512/z64, 64x64, RA flags)
CEM_FWD_GFLOP_PER_CAND_STEP = 24.46

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


def _cdiv(a, b):
    return (a + b - 1) // b


def pmc_traffic(tag, kernel_substr, workgroups):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC profile
    (profiles/*_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        rows = json.load(open(files[-1])).get(tag, [])
    except (OSError, ValueError):
        return None, None
    for r in rows:
        if kernel_substr in r["kernel"] and r["workgroups"] == workgroups:
            return r["hbm_side_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def gate_roofline(k):
    """Roofline entry of the profiled gate GEMM: exact-fp32 MFMA (157.3 TF) or the split-precision kernel
    (3 bf16 parts per operand, 6 part-products per fp32 product -> 2500 / 6 TF algorithmic peak)."""
    if k["split"]:
        peak = BF16_MFMA_PEAK_TFLOPS / 6
        return {"dtype": "bf16x3-split (fp32-equivalent, 6 bf16 MFMA products per fp32 product)", "peak": peak,
                "frac": k["tflops"] / peak, "bf16_mfma_issue_frac": 6 * k["tflops"] / BF16_MFMA_PEAK_TFLOPS}
    return {"dtype": "fp32", "peak": F32_MFMA_PEAK_TFLOPS, "frac": k["tflops"] / F32_MFMA_PEAK_TFLOPS}


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


def bench_train(args, dev, rank, world, distributed):
    cf = namespace(dev, lstm_group_norm=args.group_norm)
    if args.h48:  # the reference's default frame size (config/__init__.py:166-171): 48 x 64 -> 6 x 8 latent maps
        cf.image_height = 48
    if args.cfg5:  # BASELINE configs[4] per GPU: 128x128 frames, 8 samples, 10 predicted frames (16x16 latent maps)
        cf.image_width = cf.image_height = 128
        cf.batch_size, cf.n_future = 8, 10
    log("building trainer (g512/z64, 238.6 M params)")
    tr = PredictionTrainer(cf)
    tr.model.train()
    B, T = cf.batch_size, cf.n_past + cf.n_future
    batches = [syn.synth_video(seed=100 + rank * 1000 + i, T=T, B=B, H=cf.image_height, W=cf.image_width)
               for i in range(2)]
    batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
    for i in range(3):  # prime the caching allocator (new tensor sizes cost a hipMalloc + sync each); not a warmup step
        tr._train_step(batches[i % 2])
    torch.cuda.synchronize()
    log("allocator primed")
    for i in range(args.warmup):
        tr._train_step(batches[i % 2])
        torch.cuda.synchronize()
        log(f"train warmup step {i} done")
    g = cf.g_dim
    # dominant kernel: ConvLSTM layer-0 gate GEMM, FWD  (M = B*64, N = 4g, K = 25 * 2g)
    prof = {"match": (ops.FWD, 5, 2 * g, 4 * g), "events": []}
    ops.PROFILE = prof
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr._train_step(batches[i % 2])
    barrier_sync(distributed)
    dt = max_over_ranks(time.perf_counter() - t0, dev, distributed)
    log(f"train: {args.steps} steps in {dt:.3f} s")
    ops.PROFILE = None
    frames = world * B * T * args.steps
    fwd_gflop = 145.03 if args.cfg5 else TRAIN_FWD_GFLOP_PER_SAMPLE_STEP  # 128x128 / 64x64 train forward
    if args.h48:
        fwd_gflop *= 48.0 / 64.0  # every conv's FLOPs scale with the pixel count
    step_flop = 3 * B * (T - 1) * fwd_gflop * 1e9
    kern = profile_summary(prof, 2.0 * (4 * g) * (25 * 2 * g))
    return {"frames_per_s": frames / dt, "ms_per_step": dt / args.steps * 1e3,
            "step_tflops_per_gpu": step_flop / (dt / args.steps) / 1e12, "kernel": kern,
            "global_batch": world * B, "state": tr}


def bench_cem(args, dev, rank, world, distributed, model=None):
    n_per_gpu, horizon = args.cem_candidates, 15
    cf = namespace(dev, model_use_mask=False, model_use_future_mask=False, model_use_robot_state=False,
                   reconstruction_loss="l1", candidates_batch_size=args.cem_batch, batch_size=args.cem_batch,
                   lstm_group_norm=args.group_norm)
    model = SVGConvModel(cf)
    if distributed:
        dist.broadcast(model.flat_parameters()[0], src=0)
    model.eval()
    N = n_per_gpu * world
    prob = syn.synth_cem_problem(seed=0, N=N, T=horizon - 1)
    pol = CEMPolicy(cf, model, horizon=horizon, opt_iter=10, action_candidates=N, topk=5, init_std=0.015)
    start = State(img=prob["start_img"])
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
    g = cf.g_dim
    log(f"cem: model built, {N} candidates")
    small = syn.synth_cem_problem(seed=0, N=N, T=2)
    pol.traj_sampler.generate_model_rollouts(small["actions"], start, goal)  # allocator priming, 2 model steps
    for _ in range(args.cem_warmup):
        pol.traj_sampler.generate_model_rollouts(prob["actions"], start, goal)
        log("cem warmup iteration done")
    prof = {"match": (ops.FWD, 5, 2 * g, 4 * g), "events": []}
    ops.PROFILE = prof
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for _ in range(args.cem_iters):
        ro = pol.traj_sampler.generate_model_rollouts(prob["actions"], start, goal)
    barrier_sync(distributed)
    dt = max_over_ranks(time.perf_counter() - t0, dev, distributed)
    log(f"cem: {args.cem_iters} iterations in {dt:.3f} s")
    ops.PROFILE = None
    assert len(ro["sum_cost"]) == N and np.all(np.isfinite(ro["sum_cost"]))
    # secondary metric: the whole planner call -- sampling, rollouts, cost gather, top-k, refit --
    # at a reduced iteration count (the per-iteration cost does not depend on it)
    pol.optimization_iter = 2
    barrier_sync(distributed)
    t1 = time.perf_counter()
    pol.get_action(start, goal, 0, 0)
    barrier_sync(distributed)
    dt_ga = max_over_ranks(time.perf_counter() - t1, dev, distributed)
    it_flop_per_gpu = n_per_gpu * (horizon - 1) * CEM_FWD_GFLOP_PER_CAND_STEP * 1e9
    kern = profile_summary(prof, 2.0 * (4 * g) * (25 * 2 * g))
    return {"rollouts_per_s": N * args.cem_iters / dt, "s_per_iter": dt / args.cem_iters,
            "tflops_per_gpu": it_flop_per_gpu / (dt / args.cem_iters) / 1e12, "kernel": kern, "candidates": N,
            "candidates_batch_size": args.cem_batch,
            "get_action": {"value": N * 2 / dt_ga, "unit": "candidate-rollouts/s", "opt_iter": 2,
                           "note": "CEMPolicy.get_action end to end (sampling, rollouts, cost gather, top-k, refit)"}}


def cpu_baseline(train_state_dict, args):
    """The oracle (CPU restatement of the reference path, pinned by tests/golden) on this host's cores:
    a bounded sample of the same workloads."""
    from oracle import svg_oracle as orc
    torch.set_num_threads(host_threads())
    cores = torch.get_num_threads()
    log(f"cpu baseline on {cores} threads")
    Bs = 4
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=Bs, n_past=1, n_future=5, lr=1e-4, **RA)
    sd = {k: v.detach().cpu().clone().contiguous() for k, v in train_state_dict.items()}
    ts = orc.TrainState.create(cfg, sd)
    data = syn.synth_video(seed=100, T=6, B=Bs)
    t0 = time.perf_counter()
    orc.train_step(ts, data)
    t_train = time.perf_counter() - t0
    log(f"cpu baseline train step: {t_train:.1f} s")
    ccfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=8, candidates_batch_size=8, sample_mean=True, reward_type="dense",
                   model_use_mask=False, model_use_future_mask=False, model_use_robot_state=False,
                   reconstruction_loss="l1")
    csd = orc.make_weights(ccfg, seed=0)
    prob = syn.synth_cem_problem(seed=0, N=8, T=14)
    t0 = time.perf_counter()
    orc.cem_rollouts(csd, ccfg, prob["actions"], prob["start_img"], prob["goal_imgs"], prob["goal_masks"])
    t_cem = time.perf_counter() - t0
    return {"value": Bs * 6 / t_train, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"train: 1 step of configs[1] at batch {Bs} of 16 ({t_train:.1f} s); "
                      f"cem: 8 candidates x 14 steps, batch 8 ({t_cem:.1f} s)",
            "cem_value": 8 / t_cem, "cem_unit": "candidate-rollouts/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="both", choices=["both", "train", "cem"])
    ap.add_argument("--cem-candidates", type=int, default=1000, help="candidates per GPU")
    ap.add_argument("--cem-batch", type=int, default=1000, help="candidates per GPU pass (1000: 667 vs 661 rollouts/s at 500)")
    ap.add_argument("--cem-iters", type=int, default=2)
    ap.add_argument("--cem-warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h48", action="store_true",
                    help="train workload on 48x64 frames, the reference's default --image_height (not the headline)")
    ap.add_argument("--cfg5", action="store_true",
                    help="train workload at BASELINE configs[4] per-GPU size (128x128, bs 8, n_future 10); not the headline")
    ap.add_argument("--group-norm", action="store_true",
                    help="both workloads with --lstm_group_norm True (NormConvLSTMCell; not the headline config)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    distributed = world > 1
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    backend = os.environ.get("RAC_DIST_BACKEND", "nccl")  # "gloo" + RAC_BENCH_ONE_GPU=1: rehearse N ranks on one GPU
    if os.environ.get("RAC_BENCH_ONE_GPU") == "1":
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(1234)

    out = {"metric": "SVG train frames/sec + CEM candidate-rollouts/sec, 64x64", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "fp32", "data": "synthetic"}
    train = cem = None
    if args.workload in ("both", "train"):
        train = bench_train(args, dev, rank, world, distributed)
    if args.workload in ("both", "cem"):
        if train is not None:
            sd_keep = {k: v.detach().cpu() for k, v in train["state"].model.state_dict().items()} \
                if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
            del train["state"]
            torch.cuda.empty_cache()
        cem = bench_cem(args, dev, rank, world, distributed)
    elif train is not None:
        sd_keep = {k: v.detach().cpu() for k, v in train["state"].model.state_dict().items()} \
            if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        del train["state"]

    if train is not None:
        out.update(value=train["frames_per_s"], unit="frames/s", ms_per_step=train["ms_per_step"],
                   config={"workload": "SVG train step, BASELINE configs[1]: 64x64, bs 16/GPU, n_past 1, n_future 5, "
                                       "g_dim 512, z_dim 64, robot-aware flags (dontcare_l1)",
                           "global_batch": train["global_batch"], "parallelism": f"ddp{world}",
                           "algorithmic_tflop_per_step_per_gpu": 8.70})
        k = train["kernel"]
        if k is None:  # no launch of the profiled shape (e.g. --group-norm: separate ih / hh gate convs)
            k = {"split": False, "tflops": train["step_tflops_per_gpu"], "avg_ms": None, "launches": 0}
        gr = gate_roofline(k)
        kname = "igemm_split_bdirect16_kernel" if k["split"] else "igemm_fast_kernel<0, 128, 128"
        # PMC pass over exactly this launch (tools/bench_gemm.py at the same shape; tools/run_profiles.sh)
        traffic, traffic_src = pmc_traffic("gemm_train", kname, 512)
        out["roofline"] = {"bound": "mfma", "kernel": "FWD 5x5 ConvLSTM gate GEMM (M=1024,N=2048,K=25600), " + gr["dtype"],
                           "achieved": k["tflops"], "peak": gr["peak"], "unit": "TFLOP/s", "frac": gr["frac"],
                           "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_ms": k["avg_ms"], "launches": k["launches"],
                           "step_achieved": train["step_tflops_per_gpu"], "step_peak": F32_MFMA_PEAK_TFLOPS,
                           "step_frac": train["step_tflops_per_gpu"] / F32_MFMA_PEAK_TFLOPS,
                           "note": "step_* = algorithmic 8.70 TFLOP per step over wall time, against the exact-fp32 "
                                   "MFMA peak (the weight-gradient GEMMs and the vgg layers run there)"}
        if "bf16_mfma_issue_frac" in gr:
            out["roofline"]["bf16_mfma_issue_frac"] = gr["bf16_mfma_issue_frac"]
    if cem is not None:
        k = cem["kernel"]
        if k is None:  # no launch of the profiled shape (--group-norm: separate ih / hh gate convs)
            k = {"split": False, "tflops": cem["tflops_per_gpu"], "avg_ms": None, "launches": 0}
        gate = {"avg_launch_ms": k["avg_ms"], "tflops": k["tflops"], "launches": k["launches"]}
        gate.update(gate_roofline(k))
        gate["traffic"], gate["traffic_source"] = pmc_traffic(
            "gemm_cem", "igemm_split_bdirect16_kernel" if k["split"] else "igemm_fast_kernel<0, 128, 128",
            _cdiv(cem["candidates_batch_size"] * 64, 128) * 16)
        cem_obj = {"value": cem["rollouts_per_s"], "unit": "candidate-rollouts/s", "s_per_iteration": cem["s_per_iter"],
                   "config": {"workload": "CEM rollouts, BASELINE configs[2]: 1000 candidates/GPU x horizon 15 "
                                          "(14 model steps), frozen g512/z64 model, 64x64, dense image cost",
                              "candidates": cem["candidates"], "candidates_batch_size": cem["candidates_batch_size"],
                              "parallelism": f"candidate-shard{world}"},
                   "achieved_tflops_per_gpu": cem["tflops_per_gpu"],
                   "frac_of_f32_mfma_peak": cem["tflops_per_gpu"] / F32_MFMA_PEAK_TFLOPS,
                   "gate_gemm": gate, "get_action": cem["get_action"]}
        if train is None:
            out.update(value=cem["rollouts_per_s"], unit="candidate-rollouts/s", ms_per_step=cem["s_per_iter"] * 1e3,
                       config=cem_obj["config"],
                       roofline={"bound": "mfma", "kernel": "igemm FWD 5x5 ConvLSTM gate GEMM", "achieved": k["tflops"],
                                 "peak": gate["peak"], "unit": "TFLOP/s", "frac": gate["frac"], "traffic": gate["traffic"],
                                 "dtype": gate["dtype"]})
        out["cem"] = cem_obj
    if rank == 0 and world == 1 and not args.no_cpu_baseline and train is not None:
        out["cpu_baseline"] = cpu_baseline(sd_keep, args)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

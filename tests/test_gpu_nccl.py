"""RCCL has executed every collective call site of the path once: the box has ONE MI355X and RCCL
refuses two ranks on a device, so a fresh child process runs a one-rank "nccl" process group with RAC_DIST_FORCE=1
(robot_aware_control_amd/parallel_env.py) and drives the planner's candidate broadcast and cost all-gather, the
trainer's parameter broadcast and `GradReducer`, the sharded optimiser's reduce-scatter / all-gather -- each compared
with the same call with the collectives off -- and `bench.py --gpus 1` goes through `torch.distributed.run` with the
same switch (barrier, max-over-ranks all-reduce, per-rank all-gather, weight broadcast on RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**kw):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "RAC_DIST_BACKEND", "RAC_BENCH_ONE_GPU"):
        env.pop(k, None)
    return env


def test_every_collective_call_site_runs_on_rccl():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = _env(MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_world1_child.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl"
    assert out["gather_on_device"] and out["cem_equal"], out          # cost all-gather on device tensors
    assert out["broadcast_on_device"] and out["action_equal"], out    # candidate broadcast, one per CEM iteration
    assert out["param_broadcast"], out
    assert out["allreduce_slices"] >= 7, out                          # 6 ConvLSTM weight slices + the rest, async
    assert out["grad_rel_diff"] < 1e-6 and out["param_rel_diff"] < 1e-6 and out["loss_equal"], out
    assert out["reduce_scatter"] >= 2 and out["param_allgather"] >= 2, out
    # sharded step == all-reduce + full Adam up to what two identical plain runs differ by (atomics' order; Adam turns
    # gradient noise into +-lr steps): a small fraction of one optimiser step
    assert out["sharded_param_rel_diff"] <= max(5.0 * out["plain_rerun_rel_diff"], 1e-6) or \
        out["sharded_param_rel_diff"] < 0.01 * out["two_step_update_rel"], out
    assert out["gate_armed"] and out["staged_wait"] and out["gate_released"], out  # all-gather waited in two stages


def test_bench_under_torchrun_one_rank_rccl():
    """`bench.py --gpus 1` launched as the driver launches N > 1 (`python -m torch.distributed.run`), RCCL process group
    of one rank: the barrier / max-over-ranks / per-rank gathers, the weight broadcasts, the gradient all-reduce and the
    cost all-gather of the benchmark all run on RCCL.  A fresh process tree."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--cem-candidates", "64", "--cem-batch", "64", "--cem-iters", "1", "--cem-opt-iter", "2",
           "--exact-steps", "1", "--side-steps", "1", "--no-cem-ra", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=_env(RAC_DIST_FORCE="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["cem"]["value"] > 0
    assert "allreduce_exposed" in out["time_breakdown_ms"]            # the DDP branch of the train step ran
    assert out["cem"]["ranks"]["cost_allgather_ms"] is not None       # and the planner's all-gather
    # the exact-fp32 and configs[4] legs under a process group (the two-rank rehearsal of test_gpu_dist.py skips them), and
    # both gradient-exchange modes on RCCL
    assert out["fp32_exact"]["train"]["ms_per_step"] > 0 and out["fp32_exact"]["cem"]["rollouts_per_s"] > 0
    assert out["side"]["cfg5"]["value"] > 0 and "128x128" in out["side"]["cfg5"]["config"]["workload"]
    assert out["ddp_modes"]["sharded"].get("ms_per_step", 0) > 0, out["ddp_modes"]
    assert out["ddp_modes"]["sharded"]["optimizer"] == "ShardedAdam"
    # the side lines of the paths the reference's own command lines take: scheduled sampling, the deployed GroupNorm model
    for key in ("sched", "sched_mix", "deployed_train", "deployed_train_sched"):
        assert out["side"][key]["value"] > 0 and 0 < out["side"][key]["step_frac"] < 1, key
    assert out["side"]["deployed_cem"]["value"] > 0 and 0 < out["side"]["deployed_cem"]["frac_of_split_peak"] < 1
    assert out["dist"]["backend"] == "nccl" and out["dist"]["world_size"] == 1 and out["dist"]["devices"][0]["uuid"]

"""Per-step divergence of a training loop with the optimiser's late update overlapped (FusedAdam.overlap_next_forward)
against the same loop without it, and the run-to-run noise floor:  python tools/debug_overlap.py [lazy 0|1] [groups 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import svg_oracle as orc  # noqa: E402
from robot_aware_control_amd import ops, synthetic as syn  # noqa: E402
from tests.test_gpu_model import FLAGSETS, make_trainer  # noqa: E402

lazy = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
os.environ["RAC_ADAM_LATE_GROUPS"] = sys.argv[2] if len(sys.argv) > 2 else "0"
ops.LAZY_ZERO_GRAD = lazy
dev = torch.device("cuda:0")
cfg = orc.Cfg(g_dim=128, z_dim=16, batch_size=4, n_past=1, n_future=4, lr=1e-3, **FLAGSETS["ra"])
sd = orc.make_weights(cfg, seed=3, randomize_bn_stats=False)
pattern = os.environ.get("PATTERN", "tststs")  # t: teacher-forced window, s: scheduled-sampling window


def run(overlap):
    tr = make_trainer(cfg, sd, dev, log_dir="/tmp/rac_dbg", n_eval=5, test_batch_size=4)
    tr.optimizer.overlap_next_forward = overlap
    gen = torch.Generator().manual_seed(17)
    tr.model.eps_source = lambda shape: torch.randn(shape, generator=gen)
    snaps = []
    for step, kind in enumerate(pattern):
        data = syn.synth_video(seed=50 + step, T=5, B=4)
        losses = tr._train_step(data, use_truth=[True, True, False, True, False] if kind == "s" else None)
        late = ops.PARAM_GATE is tr.optimizer
        tr.optimizer.wait_params()
        torch.cuda.synchronize()
        flat, grad = tr.model.flat_parameters()
        snaps.append((flat.detach().cpu().clone(), grad.detach().cpu().clone(), dict(losses), late))
    ops.PARAM_GATE = None
    return snaps, tr


def report(name, a, b, tr):
    print(f"== {name}")
    for step, ((fa, ga, la, late_a), (fb, gb, lb, late_b)) in enumerate(zip(a, b)):
        ef = float((fa.double() - fb.double()).norm() / fb.double().norm())
        eg = float((ga.double() - gb.double()).norm() / (gb.double().norm() + 1e-30))
        el = max(abs(la[k] - lb[k]) / (abs(lb[k]) + 1e-12) for k in lb)
        print(f"step {step} ({pattern[step]}): params {ef:.2e}  grads {eg:.2e}  losses {el:.2e}  late {late_a}/{late_b}")
        if ef > 1e-6:
            worst = []
            for k, p in tr.model.named_parameters():
                off, n = p._rac_off, p.numel()
                d = float((fa[off:off + n].double() - fb[off:off + n].double()).norm() / (fb[off:off + n].double().norm() + 1e-30))
                dg = float((ga[off:off + n].double() - gb[off:off + n].double()).norm() / (gb[off:off + n].double().norm() + 1e-30))
                worst.append((d, dg, k))
            for d, dg, k in sorted(worst, reverse=True)[:8]:
                print(f"      {k}: param {d:.2e} grad {dg:.2e}")
            break


ref, tr = run(False)
again, _ = run(False)
report("no overlap, run 2 vs run 1 (noise floor)", again, ref, tr)
got, _ = run(True)
report("overlap vs no overlap", got, ref, tr)

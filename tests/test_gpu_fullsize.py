"""Parity AT THE BENCHMARKED SIZES (BASELINE.json configs[1], configs[2], the per-GPU shard of configs[4]) and the
tested model of why full-model gradients of two fp32 implementations differ at all (LeakyReLU / max-pool selections
that flip at pre-activations which are zero to rounding; tests/selection_tools.py).

bench.py times exactly these launch geometries: M = 1024 gate GEMMs with the XCD-grouped launch and the time-batched
weight gradient over 5 steps (configs[1]); M = 64 000 gate GEMMs, 8 000-workgroup grids (configs[2]); 16x16 latent maps
on the image-rows kernel at 1024 -> 2048 channels (configs[4])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import svg_oracle as orc  # noqa: E402
from robot_aware_control_amd import synthetic as syn  # noqa: E402
from tests.selection_tools import gpu_selections, grad_errors  # noqa: E402
from tests.test_gpu_model import FLAGSETS, FakeRobotModel, build_model, make_trainer, ns_for  # noqa: E402

# with the selections agreed, what is left is fp32 rounding through ~60 layers of BPTT
FORCED_GRAD_TOL = 1e-4   # norm-wise, per parameter
FLIP_RATE_MAX = 2e-5     # LeakyReLU / pooling selections that differ, per element (measured: a few 1e-6)
FLIP_PRE_MAX = 1e-4      # |pre-activation| / layer rms at any differing selection: "zero to rounding"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def forced_step(tr, ts, data, eps, B, update=False):
    """One GPU train step with its selections recorded, then the oracle's step with those selections forced."""
    queue = [e for pair in eps for e in pair]
    tr.model.eps_source = lambda shape: queue.pop(0)
    with gpu_selections(tr.model, tr) as sel:
        got = tr._train_step(data)
    assert not queue
    orc.FORCING = forcing = orc.Forcing(masks=sel["masks"], pools=sel["pools"], rows_per_call=B, record=True,
                                         l1_signs=sel["l1_signs"])
    try:
        ref = orc.train_step(ts, data, eps, None, do_update=update)
    finally:
        orc.FORCING = None
    return got, ref, forcing.flips


def check_flips(flips):
    n = sum(v[0] for v in flips.values())
    total = sum(v[1] for v in flips.values())
    worst = max(v[2] for v in flips.values())
    assert n <= max(4, FLIP_RATE_MAX * total), (n, total)
    assert worst < FLIP_PRE_MAX, {k: v for k, v in flips.items() if v[2] >= FLIP_PRE_MAX}
    return n, total, worst


def check_grads(tr, ts, tol=FORCED_GRAD_TOL):
    rows, cos = grad_errors(tr.model, ts)
    bad = [(e, k) for e, k in rows if e >= tol]
    assert not bad, (len(bad), len(rows), sorted(bad)[-6:], sorted(rows)[:3])
    assert cos > 1 - 1e-8, cos
    return max(e for e, _ in rows)


def test_train_step_cfg2_full_size(dev):
    """BASELINE configs[1] exactly as benchmarked: bs 16, n_past 1, n_future 5, g 512 / z 64, robot-aware flags.
    Losses <= 1e-4 (north_star tolerance); EVERY parameter's gradient <= 1e-4 norm-wise with the selections agreed."""
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=16, n_past=1, n_future=5, lr=1e-4, **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=21, randomize_bn_stats=False)
    data = syn.synth_video(seed=22, T=6, B=16)
    eps = syn.synth_eps(seed=23, steps=5, B=16, z=64, h=8, w=8)
    tr = make_trainer(cfg, sd, dev)
    tr.optimizer.step = lambda: None  # compare raw gradients
    ts = orc.TrainState.create(cfg, sd)
    got, ref, flips = forced_step(tr, ts, data, eps, 16)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, err_msg=k)
    n, total, worst = check_flips(flips)
    worst_grad = check_grads(tr, ts)
    print(f"cfg2 full size: {n} of {total} selections differ (max |pre|/rms {worst:.1e}); "
          f"worst per-parameter gradient error {worst_grad:.1e}")


def test_train_step_cfg5_shard(dev):
    """The per-GPU geometry of BASELINE configs[4] at the benchmarked width: 128x128 frames, g 512 / z 64 (16x16
    ConvLSTM maps on the image-rows kernel, k = 5 halo, 1024 -> 2048 channels), B = 2, two predicted frames."""
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, n_past=1, n_future=2, lr=1e-4, image_height=128, image_width=128,
                  **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=24, randomize_bn_stats=False)
    data = syn.synth_video(seed=25, T=3, B=2, H=128, W=128)
    eps = syn.synth_eps(seed=26, steps=2, B=2, z=64, h=16, w=16)
    tr = make_trainer(cfg, sd, dev)
    tr.optimizer.step = lambda: None
    ts = orc.TrainState.create(cfg, sd)
    got, ref, flips = forced_step(tr, ts, data, eps, 2)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, err_msg=k)
    n, total, worst = check_flips(flips)
    worst_grad = check_grads(tr, ts)
    print(f"cfg5 shard: {n} of {total} selections differ (max |pre|/rms {worst:.1e}); worst gradient error {worst_grad:.1e}")


def test_train_step_cfg5_per_gpu_size(dev):
    """BASELINE configs[4] exactly as one GPU of the 8 sees it and as `bench.py --cfg5` times it: 128x128 frames, 8
    samples, n_past 1 + n_future 10 (a 10-step BPTT window on 16x16 latent maps), g 512 / z 64, robot-aware flags.
    Losses <= 1e-4; every parameter's gradient <= 1e-4 norm-wise with the selections agreed."""
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=8, n_past=1, n_future=10, lr=1e-4, image_height=128, image_width=128,
                  **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=27, randomize_bn_stats=False)
    data = syn.synth_video(seed=28, T=11, B=8, H=128, W=128)
    eps = syn.synth_eps(seed=29, steps=10, B=8, z=64, h=16, w=16)
    tr = make_trainer(cfg, sd, dev)
    tr.optimizer.step = lambda: None
    ts = orc.TrainState.create(cfg, sd)
    got, ref, flips = forced_step(tr, ts, data, eps, 8)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, err_msg=k)
    n, total, worst = check_flips(flips)
    worst_grad = check_grads(tr, ts)
    print(f"cfg5 per-GPU size: {n} of {total} selections differ (max |pre|/rms {worst:.1e}); worst gradient error {worst_grad:.1e}")


def test_train_step_group_norm_g256_forced(dev):
    """`--lstm_group_norm True` at the width of the authors' deployed checkpoints (g 256 / z 64,
    evaluate_checkpoint.py:40-46): NormConvLSTMCell -- separate ih / hh gate convs, GroupNorm(16) on both and on the cell
    state -- through a 3-step BPTT window with the selections agreed: losses <= 1e-4, EVERY parameter's gradient
    (GroupNorm affines included) <= 1e-4 norm-wise."""
    cfg = orc.Cfg(g_dim=256, z_dim=64, batch_size=8, n_past=1, n_future=3, lr=1e-4, lstm_group_norm=True, **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=31, randomize_bn_stats=False)
    assert any("c_norm" in k for k in sd)
    data = syn.synth_video(seed=32, T=4, B=8)
    eps = syn.synth_eps(seed=33, steps=3, B=8, z=64, h=8, w=8)
    tr = make_trainer(cfg, sd, dev)
    tr.optimizer.step = lambda: None
    ts = orc.TrainState.create(cfg, sd)
    got, ref, flips = forced_step(tr, ts, data, eps, 8)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, err_msg=k)
    n, total, worst = check_flips(flips)
    worst_grad = check_grads(tr, ts)
    print(f"group norm g256: {n} of {total} selections differ (max |pre|/rms {worst:.1e}); worst gradient error {worst_grad:.1e}")


@pytest.mark.parametrize("tag", ["vanilla", "ra", "ra_sched"])
def test_three_optimizer_steps_forced(dev, tag):
    """Three Adam steps at cfg1 size with the oracle held to the GPU pass's selections at every step: the loss
    trajectory stays within 1e-4 at steps 1 and 2 (not only step 0), BatchNorm running statistics and the updated
    weights within 1e-4 / a small fraction of lr.  (The same three steps are pinned against the REAL reference's trace,
    flips included, in test_gpu_model.py::test_train_steps_vs_reference_golden.)"""
    sched = tag.endswith("_sched")
    cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, n_past=1, n_future=2, lr=1e-4, **FLAGSETS[tag.split("_")[0]])
    sd = orc.make_weights(cfg, seed=1, randomize_bn_stats=False)
    tr = make_trainer(cfg, sd, dev)
    ts = orc.TrainState.create(cfg, sd)
    keys = [k for k, _, kind in orc.param_spec(cfg) if kind != "bn_nbt"]
    total_flips = 0
    for step in range(3):
        data = syn.synth_video(seed=20 + step, T=3, B=2)
        eps = syn.synth_eps(seed=40 + step, steps=2, B=2, z=16, h=8, w=8)
        truth = [True, True, step != 1] if sched else None  # step 1 feeds the predicted frame back
        queue = [e for pair in eps for e in pair]
        tr.model.eps_source = lambda shape: queue.pop(0)
        with gpu_selections(tr.model, tr) as sel:
            got = tr._train_step(data, use_truth=truth)
        orc.FORCING = forcing = orc.Forcing(masks=sel["masks"], pools=sel["pools"], rows_per_call=2, record=True,
                                             l1_signs=sel["l1_signs"])
        try:
            ref = orc.train_step(ts, data, eps, truth, do_update=True)
        finally:
            orc.FORCING = None
        total_flips += check_flips(forcing.flips)[0]
        for k in ref:
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, err_msg=f"step {step} {k}")
        gsd = tr.model.state_dict()
        close = count = 0
        for k in keys:
            a, b = gsd[k].double().cpu(), ts.sd[k].detach().double()
            if "running_" in k:
                assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) < 1e-4, (step, k)
            else:
                # Adam's first steps move a weight by ~lr * g / (|g| + eps): an element whose gradient is ~1e-8 or
                # smaller (some BatchNorm biases) amplifies rounding up to a sign change, so compare in units of lr
                d = (a - b).abs()
                assert float(d.max()) <= 2.05 * cfg.lr * (step + 1), (step, k, float(d.max()))
                close, count = close + int((d <= 0.01 * cfg.lr).sum()), count + d.numel()
        assert close >= 0.999 * count, (step, close, count)  # measured: 0.9999
    print(f"three forced steps ({tag}): {total_flips} selections differed in all")


@pytest.mark.parametrize("ra", [False, True])
def test_cem_rollouts_cfg3_full_size(dev, ra):
    """BASELINE configs[2] exactly as benchmarked: 1000 candidates in ONE pass (M = 64 000 pixel rows, 8 000-workgroup
    gate GEMMs, 32 000-workgroup 64x64 layers) x 14 model steps through the frozen g 512 / z 64 model.  The oracle
    rolls out candidates [0:4], [496:500], [996:1000] (eval-mode arithmetic does not depend on the batch):
    sum_cost <= 1e-5 relative and the same ranking of the twelve."""
    from robot_aware_control_amd.state import DemoGoalState, State
    from robot_aware_control_amd.trajectory_sampler import TrajectorySampler
    flags = FLAGSETS["ra"] if ra else FLAGSETS["vanilla"]
    N, T = 1000, 14
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, candidates_batch_size=N, sample_mean=True,
                  reward_type="dontcare" if ra else "dense", topk=5, **flags)
    sd = orc.make_weights(cfg, seed=9, action_gain=200.0)
    prob = syn.synth_cem_problem(seed=6, N=N, T=T, with_robot=ra, goal_blend=0.15)
    model = build_model(cfg, sd, dev)
    sampler = TrajectorySampler(ns_for(cfg, dev), model,
                                robot_model=FakeRobotModel(prob["states"], prob["masks"]) if ra else None)
    start = State(img=prob["start_img"], state=np.zeros(5, np.float32), qpos=np.zeros(5, np.float32))
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
    ro = sampler.generate_model_rollouts(prob["actions"].clone(), start, goal)
    assert ro["sum_cost"].shape == (N,) and np.all(np.isfinite(ro["sum_cost"]))
    idx = np.r_[0:4, 496:500, 996:1000]
    sub = orc.Cfg(**{**cfg.__dict__, "candidates_batch_size": len(idx)})
    ref = orc.cem_rollouts(sd, sub, prob["actions"][idx], prob["start_img"], prob["goal_imgs"], prob["goal_masks"],
                           prob["states"][:, idx] if ra else None, prob["masks"][:, idx] if ra else None)["sum_cost"]
    got = ro["sum_cost"][idx]
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert err < 1e-5, err
    # ranking of the twelve: identical wherever neighbours in the reference ranking are further apart than the error
    order = np.argsort(-ref)
    gaps = np.abs(np.diff(ref[order])) / np.abs(ref).max()
    assert gaps.max() > 10 * err  # the fixture separates candidates (action_gain)
    resolvable = gaps > 10 * err
    got_order = np.argsort(-got)
    if resolvable.all():
        assert list(got_order) == list(order)
    for i in np.nonzero(resolvable)[0]:
        assert got[order[i]] > got[order[i + 1]]
    print(f"cfg3 full size ({'ra' if ra else 'vanilla'}): sum_cost rel err {err:.1e}, min gap {gaps.min():.1e}")


@pytest.mark.parametrize("ra", [False, True])
def test_cem_costs_do_not_depend_on_batching(dev, ra):
    """A candidate's cost must not depend on `candidates_batch_size` or on which other candidates share its pass (the
    reference's per-candidate arithmetic is batch-independent, trajectory_sampler.py:123-174): the frozen model scales
    every image by its own maximum and never splits K, so sum_cost is the same BITS for one pass of 1000, five of 200,
    ragged batches of 137, the candidates in reverse order, and a lone candidate.  g 512 / z 64, 4 model steps."""
    from robot_aware_control_amd.state import DemoGoalState, State
    from robot_aware_control_amd.trajectory_sampler import TrajectorySampler
    flags = FLAGSETS["ra"] if ra else FLAGSETS["vanilla"]
    N, T = 1000, 4
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, candidates_batch_size=N, sample_mean=True,
                  reward_type="dontcare" if ra else "dense", topk=5, **flags)
    sd = orc.make_weights(cfg, seed=9, action_gain=200.0)
    prob = syn.synth_cem_problem(seed=6, N=N, T=T, with_robot=ra, goal_blend=0.15)
    model = build_model(cfg, sd, dev)
    start = State(img=prob["start_img"], state=np.zeros(5, np.float32), qpos=np.zeros(5, np.float32))
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])

    def run(per, idx=None):
        ns = ns_for(cfg, dev, candidates_batch_size=per)
        acts = prob["actions"] if idx is None else prob["actions"][idx]
        rm = None
        if ra:
            rm = FakeRobotModel(prob["states"] if idx is None else prob["states"][:, idx],
                                prob["masks"] if idx is None else prob["masks"][:, idx])
        return TrajectorySampler(ns, model, robot_model=rm).generate_model_rollouts(acts.clone(), start, goal)["sum_cost"]

    full = run(1000)
    assert np.all(np.isfinite(full)) and np.unique(full).size > 900
    for per in (200, 137):
        assert np.array_equal(run(per), full), per
    rev = np.arange(N - 1, -1, -1)
    assert np.array_equal(run(1000, rev), full[rev])
    few = np.array([977, 3, 500])
    assert np.array_equal(run(3, few), full[few])
    assert np.array_equal(run(1, few[:1]), full[few[:1]])


demo_problem = syn.demo_problem  # the well-separated-elites fixture (also what bench.py times and checks)


@pytest.mark.parametrize("ra", [False, True])
def test_cem_elite_indices_cfg3_full_size(dev, ra):
    """north_star: "CEM elite indices are bit-exact" -- at the benchmarked size.  1000 candidates x 14 steps in ONE pass
    on the GPU; the oracle then re-rolls the GPU's OWN top 32 plus 32 random others: sum_cost <= 1e-5 relative, the
    oracle's top 5 (indices AND order) equal the GPU's, and the margins that make this a statement about all 1000
    candidates are asserted: rank 5 / rank 6 gap >= 1e-3 relative, rank 5 / rank 33 gap far above the error bound."""
    from robot_aware_control_amd.state import DemoGoalState, State
    from robot_aware_control_amd.trajectory_sampler import TrajectorySampler
    flags = FLAGSETS["ra"] if ra else FLAGSETS["vanilla"]
    N, T, K = 1000, 14, 5
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, candidates_batch_size=N, sample_mean=True,
                  reward_type="dontcare" if ra else "dense", topk=K, **flags)
    sd = orc.make_weights(cfg, seed=9, action_gain=1000.0)
    prob, demo = demo_problem(ra, N, T)
    model = build_model(cfg, sd, dev)
    start = State(img=prob["start_img"], state=np.zeros(5, np.float32), qpos=np.zeros(5, np.float32))
    sl = lambda key, idx: prob[key][:, idx] if ra else None
    # the demonstration's frames (GPU rollout of candidate N) become the per-step goal images, as uint8 like a camera's
    one = TrajectorySampler(ns_for(cfg, dev), model,
                            robot_model=FakeRobotModel(sl("states", [N]), sl("masks", [N])) if ra else None)
    placeholder = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
    obs = one.generate_model_rollouts(demo[None].clone(), start, placeholder, ret_obs=True)["obs"][0]  # (T,3,H,W)
    goal_imgs = syn.frames_to_goal_images(obs)
    goal_masks = [prob["goal_masks"][0]] * T
    goal = DemoGoalState(imgs=goal_imgs, masks=goal_masks)
    cand = np.arange(N)
    sampler = TrajectorySampler(ns_for(cfg, dev), model,
                                robot_model=FakeRobotModel(sl("states", cand), sl("masks", cand)) if ra else None)
    got = sampler.generate_model_rollouts(prob["actions"][:N].clone(), start, goal)["sum_cost"]
    assert got.shape == (N,) and np.all(np.isfinite(got))
    order = np.argsort(-got, kind="stable")
    top = order[:32]
    rest = np.random.RandomState(0).choice(order[32:], 32, replace=False)
    idx = np.concatenate([top, rest])
    sub = orc.Cfg(**{**cfg.__dict__, "candidates_batch_size": len(idx)})
    ref = orc.cem_rollouts(sd, sub, prob["actions"][idx], prob["start_img"], goal_imgs, goal_masks,
                           sl("states", idx), sl("masks", idx))["sum_cost"]
    scale = np.abs(ref).max()
    abs_err = np.abs(got[idx] - ref)
    err = float(abs_err.max() / scale)                    # the other tests' convention: relative to the largest cost
    err_own = float((abs_err[:32] / np.abs(ref[:32])).max())  # each of the GPU's top 32 against ITS OWN cost
    c = got[order]
    gaps = (c[:K] - c[1:K + 1]) / np.abs(c[1:K + 1])      # rank 1/2 ... rank K/K+1, relative to the worse of the pair
    far = float((c[K - 1] - c[31]) / abs(c[31]))
    print(f"cfg3 elites ({'ra' if ra else 'vanilla'}): top-{K} {list(order[:K])} costs {' '.join(f'{v:.5g}' for v in c[:K + 1])}; "
          f"sum_cost err {err:.1e} of max |cost| {scale:.4g}, {err_own:.1e} of each elite's own cost; "
          f"relative gaps {' '.join(f'{g:.1e}' for g in gaps)}; rank {K} / rank 32 gap {far:.1e}")
    assert err < 1e-5, err
    assert err_own < 1e-4, err_own            # north_star's tolerance, on costs that measure small frame differences
    assert gaps[K - 1] >= 1e-3, gaps          # the fixture keeps the K / K+1 gap the survey asks for
    assert (c[:K] - c[1:K + 1]).min() > 100 * abs_err[:32].max() and c[K - 1] - c[31] > 100 * abs_err.max()
    ref_order = idx[np.argsort(-ref, kind="stable")]
    assert list(ref_order[:K]) == list(order[:K]), (ref_order[:K], order[:K])
    # and through the policy's own selection (cem.py:96-97): torch.topk over the full cost vector
    import torch as _t
    assert list(_t.from_numpy(got).topk(K)[1].numpy()) == list(order[:K])


@pytest.mark.parametrize("ra", [False, True])
def test_cem_pass_in_stream_parts_is_the_same_bits(dev, ra):
    """`cfg.cem_streams = 2 / 3`: a candidate pass cut into parts that run on their own HIP streams in lock-step (one part's
    memory-bound kernels under the other's matrix-pipe kernels) gives the SAME BITS as the one-stream pass -- candidates are
    independent and the frozen model's arithmetic does not depend on the batch.  g 512 / z 64, 1000 candidates, 4 steps; also
    a ragged split (batches of 437) and a second call on the same sampler (stream reuse)."""
    from robot_aware_control_amd.state import DemoGoalState, State
    from robot_aware_control_amd.trajectory_sampler import TrajectorySampler
    flags = FLAGSETS["ra"] if ra else FLAGSETS["vanilla"]
    N, T = 1000, 4
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, candidates_batch_size=N, sample_mean=True,
                  reward_type="dontcare" if ra else "dense", topk=5, **flags)
    sd = orc.make_weights(cfg, seed=9, action_gain=200.0)
    prob = syn.synth_cem_problem(seed=6, N=N, T=T, with_robot=ra, goal_blend=0.15)
    model = build_model(cfg, sd, dev)
    start = State(img=prob["start_img"], state=np.zeros(5, np.float32), qpos=np.zeros(5, np.float32))
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])

    def sampler(streams, per):
        rm = FakeRobotModel(prob["states"], prob["masks"]) if ra else None
        return TrajectorySampler(ns_for(cfg, dev, candidates_batch_size=per, cem_streams=streams), model, robot_model=rm)
    ref = sampler(1, N).generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
    assert np.all(np.isfinite(ref)) and np.unique(ref).size > 900
    for streams, per in ((2, N), (3, N), (2, 437)):
        smp = sampler(streams, per)
        smp.generate_model_rollouts(prob["actions"][:64].clone(), start, goal)  # (a sampler's first call: one stream)
        for rep in range(2):
            got = smp.generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
            assert np.array_equal(got, ref), (streams, per, rep, float(np.abs(got - ref).max()))
        assert len(smp._streams) == streams - 1

"""Robot-aware CEM inputs on the device (robot_aware_control_amd/robot_atlas.py, rac_cem_robot_inputs): states against
the reference's own predict_batch (golden vectors, oracle/gen_golden.py:gen_robot_states), masks against the renderer
the atlas was built from, and a robot-aware get_action with no per-candidate Python in the loop."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import svg_oracle as orc  # noqa: E402
from robot_aware_control_amd import synthetic as syn  # noqa: E402
from robot_aware_control_amd.robot_atlas import (LOCO_WX250S_DIFF, WORKSPACE_HIGH, WORKSPACE_LOW,  # noqa: E402
                                                 AtlasRobotModel)

H, W = 64, 64
LOW, HIGH = torch.tensor([WORKSPACE_LOW]), torch.tensor([WORKSPACE_HIGH])


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class ToyArm:
    """Stands in for the analytical robot model (CPU IK + MuJoCo render): per-candidate Python loop, states by the
    reference's recipe, mask = an "arm" drawn from the image's bottom centre to the projected end effector."""
    push_height = 0.12
    calls = 0

    def render(self, x, y):
        u, v = (y + 0.3) / 0.6 * (W - 1), (0.55 - x) / 0.535 * (H - 1)  # a fixed top-down camera
        yy, xx = np.mgrid[0:H, 0:W]
        m = (xx - u) ** 2 + (yy - v) ** 2 <= 36
        for s in np.linspace(0, 1, 24):
            m |= (xx - ((W - 1) / 2 * (1 - s) + u * s)) ** 2 + (yy - ((H - 1) * (1 - s) + v * s)) ** 2 <= 9
        return m

    def predict_batch(self, data, thick=True):
        T1, N, _ = data["states"].shape
        states = torch.zeros((T1, N, 5))
        masks = torch.zeros((T1, N, 1, H, W))
        for i in range(N):  # the loop the atlas removes
            ToyArm.calls += 1
            low, high = data["low"][i], data["high"][i]
            raw = data["states"][0, i] * (high - low) + low
            seq = [raw.clone()]
            xy = raw[:2].double()
            for t in range(T1 - 1):
                xy = xy + data["actions"][t, i, :2].double()
                seq.append(torch.tensor([xy[0], xy[1], self.push_height, 0, 0], dtype=torch.float32))
            raw_seq = torch.stack(seq)
            states[:, i] = (raw_seq - low) / (high - low)
            for t in range(T1):
                masks[t, i, 0] = torch.from_numpy(self.render(float(raw_seq[t, 0]), float(raw_seq[t, 1])))
        return states, masks


def test_states_vs_reference_golden(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "robot_states.npz"))
    T, N = g["actions"].shape[0], g["actions"].shape[1]
    atlas = torch.zeros((2, 2, 48, 64), dtype=torch.uint8)
    for tag, diff in (("wx250s", LOCO_WX250S_DIFF), ("locobot", (0.0, 0.0))):
        rm = AtlasRobotModel(atlas, 0.0, -0.3, 0.5, 0.6, float(g[f"{tag}_push_height"]), diff, dev)
        states = torch.zeros((T + 1, N, 5))
        states[0] = torch.from_numpy(g["start"])
        out, masks = rm.predict_batch({"states": states, "actions": torch.from_numpy(g["actions"]),
                                       "low": torch.from_numpy(g["low"]).repeat(N, 1),
                                       "high": torch.from_numpy(g["high"]).repeat(N, 1)})
        assert out.shape == (T + 1, N, 5) and masks.shape == (T + 1, N, 1, 48, 64)
        assert float((out.cpu() - torch.from_numpy(g[f"{tag}_states"])).abs().max()) < 2e-7, tag


@pytest.fixture(scope="module")
def atlas_model(dev):
    arm = ToyArm()
    rm = AtlasRobotModel.build(arm, np.zeros(5), (0.05, 0.50), (-0.25, 0.25), 91, 101, arm.push_height, device=dev)
    assert rm.atlas.shape == (101, 91, H, W) and abs(rm.dx - 0.005) < 1e-9 and abs(rm.dy - 0.005) < 1e-9
    return arm, rm


def test_masks_vs_renderer(dev, atlas_model, tmp_path):
    arm, rm = atlas_model
    path = str(tmp_path / "atlas.npz")
    rm.save(path)
    rm2 = AtlasRobotModel.load(path, dev)
    assert torch.equal(rm2.atlas, rm.atlas) and (rm2.x0, rm2.dx, rm2.push_height) == (rm.x0, rm.dx, rm.push_height)
    N, T = 24, 6
    g = np.random.Generator(np.random.Philox(key=[3, 3]))
    start_xy = np.array([0.30, 0.0])
    start = (torch.tensor([[start_xy[0], start_xy[1], 0.2, 0.3, 0.0]]) - LOW) / (HIGH - LOW)
    data = {"states": torch.zeros((T + 1, N, 5)), "low": LOW.repeat(N, 1), "high": HIGH.repeat(N, 1)}
    data["states"][0] = start
    # (a) actions that are whole grid steps: the end effector sits on atlas nodes -> the masks are the renderer's
    acts = torch.zeros((T, N, 5))
    acts[:, :, :2] = torch.from_numpy(g.integers(-8, 9, (T, N, 2)) * 0.005).float()
    data["actions"] = acts
    s_ref, m_ref = arm.predict_batch(data)
    s_got, m_got = rm.predict_batch(data)
    assert float((s_got.cpu() - s_ref).abs().max()) < 1e-6
    assert torch.equal(m_got.cpu(), m_ref)
    # (b) arbitrary clamped actions: nearest node, off by at most half a grid step = a fraction of a pixel
    acts[:, :, :2] = torch.from_numpy(np.clip(g.standard_normal((T, N, 2)) * 0.03, -0.05, 0.05)).float()
    s_ref, m_ref = arm.predict_batch(data)
    s_got, m_got = rm.predict_batch(data)
    assert float((s_got.cpu() - s_ref).abs().max()) < 1e-6
    diff = (m_got.cpu() != m_ref).float().mean((2, 3, 4))
    assert float(diff.max()) < 0.01 and float(diff.mean()) < 0.003  # < 1 % of the pixels, at the silhouette's edge


def test_robot_aware_get_action_without_per_candidate_python(dev, atlas_model):
    """CEMPolicy.get_action with the robot-aware flags: the atlas model is called once per CEM iteration for all
    candidates (the toy arm: once per candidate), and plans the same action."""
    from robot_aware_control_amd.cem import CEMPolicy
    from robot_aware_control_amd.model import SVGConvModel
    from robot_aware_control_amd.state import DemoGoalState, State
    arm, rm = atlas_model
    flags = dict(model_use_mask=True, model_use_future_mask=True, model_use_robot_state=True, reconstruction_loss="dontcare_l1")
    cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, candidates_batch_size=16, sample_mean=True, reward_type="dontcare",
                  topk=3, **flags)
    d = dict(cfg.__dict__)
    d.update(device=dev, debug_cem=False, log_dir="/tmp/rac_test", img_cost_threshold=None, img_cost_world_norm=True,
             experiment="control_toy", robot_joint_dim=5, cem_shard=True)
    ns = argparse.Namespace(**d)
    model = SVGConvModel(ns)
    model.load_state_dict({k: v.clone() for k, v in orc.make_weights(cfg, seed=9, action_gain=200.0).items()})
    model.eval()
    N, T = 32, 4
    prob = syn.synth_cem_problem(seed=11, N=N, T=T, goal_blend=0.15)
    start = State(img=prob["start_img"], state=np.array([0.30, 0.0, 0.2, 0.0, 0.0], np.float32), qpos=np.zeros(5, np.float32))
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=[arm.render(0.4, 0.1)[None]])
    noise = [torch.from_numpy(np.round(np.random.Generator(np.random.Philox(key=[i, 5])).standard_normal((N, T, 2)) * 2) / 2).float()
             for i in range(2)]  # with init_std 0.01: actions on the 5 mm grid -> identical masks for both models
    acts = {}
    for name, robot in (("atlas", rm), ("toy", arm)):
        pol = CEMPolicy(ns, model, horizon=T + 1, opt_iter=2, action_candidates=N, topk=3, init_std=0.01, robot_model=robot)
        pol.trace = []
        ToyArm.calls = 0
        acts[name] = pol.get_action(start, goal, 0, 0, noise=[n.clone() for n in noise])
        acts[name + "_trace"] = pol.trace
        acts[name + "_calls"] = ToyArm.calls
    assert acts["atlas_calls"] == 0 and acts["toy_calls"] == 2 * N
    # iteration 0 (candidates on the atlas grid): same costs, same elites, same refit
    a, b = acts["atlas_trace"][0], acts["toy_trace"][0]
    np.testing.assert_allclose(a["sum_cost"], b["sum_cost"], rtol=1e-5)
    assert list(a["top_idx"]) == list(b["top_idx"])
    np.testing.assert_allclose(a["mean"], b["mean"], rtol=1e-6, atol=1e-9)
    # iteration 1 samples off the grid: nearest-node masks, costs within the mask approximation
    a, b = acts["atlas_trace"][1], acts["toy_trace"][1]
    assert np.abs(a["sum_cost"] - b["sum_cost"]).max() / np.abs(b["sum_cost"]).max() < 2e-2
    assert acts["atlas"].shape == (T, 2) and np.all(np.isfinite(acts["atlas"]))


def test_atlas_masks_quantified_and_exact_elites(dev):
    """How much do nearest-grid-node masks cost?  cfg3 geometry (1000 candidates x 14 steps, g 512 / z 64, robot-aware
    flags, dontcare cost) with the synthetic arm: costs and elite sets of atlas masks at 20 / 10 / 5 / 2.5 mm grid spacing
    against EXACTLY rendered masks for every candidate -- and the remedy that does not depend on the spacing: with
    `cem_exact_elites = 20` the 20 best candidates of the atlas pass are re-rolled with exact masks, after which the
    top 5 (indices, order, and cost BITS: rollouts are batch-invariant) are the exact run's."""
    from robot_aware_control_amd.model import SVGConvModel
    from robot_aware_control_amd.state import DemoGoalState, State
    from robot_aware_control_amd.trajectory_sampler import TrajectorySampler
    flags = dict(model_use_mask=True, model_use_future_mask=True, model_use_robot_state=True, reconstruction_loss="dontcare_l1")
    N, T, K = 1000, 14, 5
    cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, candidates_batch_size=N, sample_mean=True, reward_type="dontcare",
                  topk=K, **flags)
    d = dict(cfg.__dict__)
    d.update(device=dev, debug_cem=False, log_dir="/tmp/rac_test", img_cost_threshold=None, img_cost_world_norm=True,
             experiment="control_toy", robot_joint_dim=5, cem_shard=True, cem_exact_elites=0)
    ns = argparse.Namespace(**d)
    model = SVGConvModel(ns)
    model.load_state_dict(syn.synth_state_dict(model, seed=9))
    model.eval()
    prob = syn.synth_cem_problem(seed=11, N=N, T=T, goal_blend=0.15)
    arm = syn.SyntheticArmModel(dev)
    start = State(img=prob["start_img"], state=np.array([0.28, 0.0, 0.12, 0.0, 0.0], np.float32), qpos=np.zeros(5, np.float32))
    goal_mask = syn.arm_mask([0.40], [0.10]).numpy().astype(bool)
    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=[goal_mask])
    exact = TrajectorySampler(ns, model, robot_model=arm).generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
    order = np.argsort(-exact, kind="stable")
    scale = np.abs(exact).max()
    print(f"exact masks: top-{K} {list(order[:K])}, K/K+1 gap {(exact[order[K - 1]] - exact[order[K]]) / scale:.1e}")
    rows = []
    for nx, ny, mm in ((28, 31, 20), (54, 61, 10), (108, 121, 5), (215, 241, 2.5)):
        rm = arm.atlas(nx, ny)
        _, m_atlas = rm.predict_batch(TrajectorySampler(ns, model, robot_model=rm)._robot_data(prob["actions"], start, N, T))
        _, m_exact = arm.predict_batch(TrajectorySampler(ns, model, robot_model=rm)._robot_data(prob["actions"], start, N, T))
        pix = float((m_atlas != m_exact).float().mean())
        got = TrajectorySampler(ns, model, robot_model=rm).generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
        err = float(np.abs(got - exact).max() / scale)
        mean_err = float(np.abs(got - exact).mean() / scale)
        o = np.argsort(-got, kind="stable")
        same5 = len(set(o[:K]) & set(order[:K]))
        in20 = len(set(order[:K]) & set(o[:20]))
        ns.cem_exact_elites = 20
        calls = arm.calls
        ref = TrajectorySampler(ns, model, robot_model=rm).generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
        ns.cem_exact_elites = 0
        assert arm.calls == calls + 1  # ONE exact predict_batch, for the 20 finalists
        o2 = np.argsort(-ref, kind="stable")
        fixed = list(o2[:K]) == list(order[:K]) and np.array_equal(ref[o2[:K]], exact[order[:K]])
        rows.append((mm, pix, mean_err, same5, in20, fixed))
        print(f"atlas {mm} mm ({nx}x{ny} nodes): {100 * pix:.2f} % of mask pixels differ, cost error mean {mean_err:.1e} / max "
              f"{err:.1e} of max |cost|, "
              f"{same5}/{K} of the exact top-{K} in the atlas top-{K}, {in20}/{K} in its top-20; exact-elite refinement "
              f"restores the exact top-{K}: {fixed}")
    # finer grids differ in fewer pixels and cost less ON AVERAGE (the worst candidate's error is one silhouette pixel of
    # one step, whatever the spacing: only exact masks remove it); < 1 % of pixels at 5 mm
    assert rows[-1][1] < rows[0][1] and rows[-1][2] <= rows[0][2] and rows[2][1] < 0.01
    assert all(r[4] == K for r in rows[1:])                  # the exact elites survive the atlas screening (<= 10 mm)
    assert all(r[5] for r in rows[1:])                       # ... so re-rolling 20 finalists restores them, bit for bit

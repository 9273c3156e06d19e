"""GPU-side search for a cfg3 fixture (1000 candidates x 14 steps, g512) whose elite set is well separated:
prints, per (flags, action_gain, goal_blend, seed), the relative spread of sum_cost and the gaps between the nine best
candidates, so that tests/test_gpu_fullsize.py can pin a fixture with a K / K+1 gap >= 1e-3."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import svg_oracle as orc  # noqa: E402  (make_weights only: the synthetic state_dict generator)
from robot_aware_control_amd import synthetic as syn  # noqa: E402
from robot_aware_control_amd.state import DemoGoalState, State  # noqa: E402
from robot_aware_control_amd.trajectory_sampler import TrajectorySampler  # noqa: E402
from tests.test_gpu_model import FLAGSETS, FakeRobotModel, build_model, ns_for  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    N, T = 1000, 14
    for ra in (False, True):
        flags = FLAGSETS["ra"] if ra else FLAGSETS["vanilla"]
        cfg = orc.Cfg(g_dim=512, z_dim=64, batch_size=2, candidates_batch_size=N, sample_mean=True,
                      reward_type="dontcare" if ra else "dense", topk=5, **flags)
        for gain in (200.0, 1000.0):
            sd = orc.make_weights(cfg, seed=9, action_gain=gain)
            model = build_model(cfg, sd, dev)
            for blend in (0.15, 0.05):
                for seed in (6, 7, 8, 9):
                    prob = syn.synth_cem_problem(seed=seed, N=N, T=T, with_robot=ra, goal_blend=blend)
                    sampler = TrajectorySampler(ns_for(cfg, dev), model,
                                                robot_model=FakeRobotModel(prob["states"], prob["masks"]) if ra else None)
                    start = State(img=prob["start_img"], state=np.zeros(5, np.float32), qpos=np.zeros(5, np.float32))
                    goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
                    c = sampler.generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
                    order = np.argsort(-c)
                    top = c[order[:9]]
                    gaps = -np.diff(top) / np.abs(c).max()
                    print(f"ra={int(ra)} gain={gain:g} blend={blend} seed={seed}: mean {c.mean():.5g} "
                          f"std/|mean| {c.std() / abs(c.mean()):.2e} top gaps " + " ".join(f"{g:.1e}" for g in gaps),
                          flush=True)
            del model
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

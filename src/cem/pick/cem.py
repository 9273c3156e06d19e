"""`from src.cem.pick.cem import CEMPolicy` (reference src/cem/pick/cem.py): 4-D actions, learned-physics branch."""
from robot_aware_control_amd.cem import SimCEMPolicy


class CEMPolicy(SimCEMPolicy):
    def __init__(self, cfg, physics="gt", horizon=5, opt_iter=10, action_candidates=100, topk=5, init_std=1.0,
                 **kw):
        super().__init__(cfg, physics, horizon, opt_iter, action_candidates, topk, init_std, action_dim=4, **kw)

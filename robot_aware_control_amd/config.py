"""Command-line flags of the hot path, name- and default-compatible with the reference's
argparse tree (reference src/config/__init__.py:23-365).  Booleans are the strings True/False.

Table-driven: (flag, type, default[, choices]).  Flags that the reference README / scripts still
pass but its parser no longer declares (`--stoch`, `--experiment singlerobot`, `--multiview`
outside FetchPush; SURVEY.md caveat C2) are accepted and ignored so published command lines run.
"""
from __future__ import annotations

import argparse


def str2bool(v):
    return v.lower() == "true"


def str2intlist(value):
    return [int(n) for n in value.split(",")] if value else value


def str2list(value):
    return value.split(",") if value else value


B, I, F, S = str2bool, int, float, str

_GENERAL = [
    ("jobname", S, None), ("log_dir", S, "logs"), ("wandb", B, False), ("wandb_entity", S, "pal"),
    ("wandb_project", S, "roboaware"), ("wandb_group", S, None), ("wandb_job_type", S, None),
    ("reward_type", S, "weighted", ["weighted", "dense", "inpaint", "sparseblackrobot", "inpaint-blur",
                                    "eef_inpaint", "dontcare"]),
    ("most_recent_background", B, False), ("blur_sigma", F, 10), ("unblur_cost_scale", F, 3),
    ("unblur_timestep", F, 1), ("mbrl_algo", S, "cem", ["cem"]), ("gpu", I, None), ("seed", I, 0),
    ("num_episodes", I, 100), ("record_trajectory", B, False), ("record_trajectory_interval", I, 5),
    ("record_video_interval", I, 1), ("env", S, "FetchPush", ["FetchPush", "LocobotTable", "LocobotPick"]),
]

_PREDICTION = [
    ("lr", F, 0.0003), ("beta1", F, 0.9), ("batch_size", I, 100), ("test_batch_size", I, 16),
    ("optimizer", S, "adam"), ("niter", I, 300), ("epoch_size", I, 600), ("image_width", I, 64),
    ("image_height", I, 48), ("channels", I, 3), ("dataset", S, "smmnist"), ("n_past", I, 1), ("n_future", I, 9),
    ("n_eval", I, 10), ("checkpoint_interval", I, 5), ("eval_interval", I, 5), ("rnn_size", I, 256),
    ("prior_rnn_layers", I, 2), ("posterior_rnn_layers", I, 2), ("predictor_rnn_layers", I, 2), ("z_dim", I, 10),
    ("g_dim", I, 128), ("action_dim", I, 2), ("action_enc_dim", I, 2), ("robot_dim", I, 6), ("robot_enc_dim", I, 6),
    ("robot_joint_dim", I, 7), ("beta", F, 0.0001), ("last_frame_skip", B, False),
    ("model", S, "svg", ["svg", "det", "copy", "cdna_det"]), ("model_use_mask", B, False),
    ("model_use_future_mask", B, False), ("model_use_robot_state", B, True),
    ("model_use_future_robot_state", B, False), ("model_use_heatmap", B, False),
    ("model_use_future_heatmap", B, False), ("black_robot_input", B, False),
    ("reconstruction_loss", S, "mse", ["mse", "l1", "dontcare_mse", "dontcare_l1"]),
    ("scheduled_sampling", B, False), ("scheduled_sampling_k", I, 4000), ("robot_pixel_weight", F, 0),
    ("learned_robot_model", B, False), ("robot_model_ckpt", S, None), ("cdna_kernel_size", I, 5),
    ("lstm_group_norm", B, False), ("sample_mean", B, False),
]

_DATASET = [
    ("data_threads", I, 5), ("data_root", S, "data"), ("train_val_split", F, 0.8), ("temporal_beta", F, 1),
    ("demo_length", I, 12), ("action_noise", F, 0),
    ("video_type", S, "object_inpaint_demo", ["object_inpaint_demo", "robot_demo", "object_only_demo"]),
    ("video_length", I, 31), ("impute_autograsp_action", B, True), ("preload_ram", B, False),
    ("experiment", S, "train_robonet",
     ["train_robonet", "train_sawyer_multiview", "finetune_sawyer_view", "finetune_widowx",
      "train_locobot_singleview", "train_locobot_table", "train_locobot_pick", "finetune_locobot", "eval_franka",
      "control_franka", "control_wx250s", "singlerobot"]),
    ("preprocess_action", S, "raw", ["raw", "camera_raw", "state_infer", "camera_state_infer"]),
    ("img_augmentation", B, False), ("color_jitter_range", F, 0.1), ("random_crop_size", I, 59),
    ("dropout", F, None), ("world_error_dict", S, None), ("finetune_num_train", I, 400),
    ("finetune_num_test", I, 100), ("random_snippet", B, True), ("load_movement_info", B, False),
    ("movement_weight", F, 1.0),
]

_CEM = [
    ("horizon", I, 5), ("opt_iter", I, 10), ("action_candidates", I, 30), ("topk", I, 5), ("replan_every", I, 1),
    ("dynamics_model_ckpt", S, None), ("candidates_batch_size", I, 200), ("use_env_dynamics", B, False),
    ("debug_trajectory_path", S, None), ("debug_cem", B, False), ("object_demo_dir", S, None),
    ("subgoal_start", I, 0), ("sequential_subgoal", B, True), ("demo_cost", B, False), ("demo_timescale", I, 1),
    ("action_repeat", I, 1),
    ("demo_type", S, "object_only_demo", ["object_inpaint_demo", "object_only_demo", "robot_demo"]),
    ("cem_init_std", F, 1), ("sparse_cost", B, False), ("cem_open_loop", B, False),
    ("cem_prediction_use_thick_mask", B, False),
]

_COST = [
    ("world_cost_success", F, 4000), ("robot_cost_success", F, 0.01), ("robot_cost_weight", F, 0),
    ("world_cost_weight", F, 1), ("img_cost_threshold", F, None), ("img_cost_world_norm", B, True),
    ("subgoal_completion_bonus", F, 0),
]

# env-specific groups (reference config/__init__.py:105-148); only the fields the hot path may read
_ENV = {
    "FetchPush": [("img_dim", I, 128), ("camera_name", S, "external_camera_0"), ("camera_ids", str2intlist, [0, 4]),
                  ("pixels_ob", B, True), ("norobot_pixels_ob", B, False), ("robot_mask_with_obj", B, False),
                  ("inpaint_eef", B, True), ("depth_ob", B, False), ("object_dist_threshold", F, 0.01),
                  ("gripper_dist_threshold", F, 0.025), ("push_dist", F, 0.2), ("max_episode_length", I, 10),
                  ("robot_goal_distribution", S, "random"), ("large_block", B, False), ("red_robot", B, False),
                  ("invisible_demo", B, False), ("demo_dir", S, "demos/fetch_push")],
    "LocobotTable": [("modified", B, False), ("demo_dir", S, "demos/locobot_table")],
    "LocobotPick": [("modified", B, False), ("demo_dir", S, "demos/locobot_pick"), ("cyclegan", B, False),
                    ("goal_image_type", S, "image")],
}

# stale-but-published flags: accepted, ignored
_COMPAT = [("stoch", B, False), ("multiview", B, False)]

# MI355X additions (not in the reference)
_NATIVE = [("ddp_bucket_mb", I, 64), ("cem_shard", B, True),
           ("cem_exact_elites", I, 0),  # re-roll the M best candidates of an atlas pass with exactly rendered masks
           ("cem_shared_start", B, True),  # planner step 0: encode the (shared) start frame once, not per candidate
           ("ddp_shard_optimizer", B, False),  # DDP: reduce-scatter + Adam on 1/world slices + parameter all-gather
           ("plot", B, False)]          # write the per-epoch generation GIFs of PredictionTrainer.plot


def _add(parser, table):
    for row in table:
        name, typ, default = row[:3]
        kw = {"type": typ, "default": default}
        if len(row) > 3:
            kw["choices"] = row[3]
        parser.add_argument("--" + name, **kw)


def create_parser(env: str | None = None):
    parser = argparse.ArgumentParser("Robot Aware Cost", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    _add(parser, _GENERAL)
    if env is None:
        env = parser.parse_known_args()[0].env
    for table in (_PREDICTION, _DATASET, _COST, _CEM, _ENV[env], _COMPAT, _NATIVE):
        _add(parser, table)
    if env == "FetchPush":  # reference add_fetch_push_arguments overrides these defaults
        parser.set_defaults(robot_dim=6, robot_enc_dim=6)
    return parser


def argparser(argv=None):
    """Parse; unknown flags abort exactly like the reference (config/__init__.py:360-365)."""
    parser = create_parser(None if argv is None else create_env_probe(argv))
    args, unparsed = parser.parse_known_args(argv)
    assert len(unparsed) == 0, unparsed
    return args, unparsed


def create_env_probe(argv):
    probe = argparse.ArgumentParser(add_help=False)
    probe.add_argument("--env", type=str, default="FetchPush")
    return probe.parse_known_args(argv)[0].env

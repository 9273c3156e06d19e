"""CPU: the C-ABI library loads and exports every symbol include/rac_hip.h declares; host-side logic
(flag parser, sharding, state_dict layout, checkpoint-compatible optimizer state)."""
import argparse
import os
import re

import numpy as np
import pytest
import torch

from oracle import svg_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import robot_aware_control_amd as rac
    hdr = open(os.path.join(ROOT, "include", "rac_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(rac_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 30
    assert declared == set(rac.EXPORTS), declared ^ set(rac.EXPORTS)
    lib = rac.load()  # dlopen + every symbol typed; raises on a missing one
    assert lib.rac_version() == 11
    assert lib.rac_device_arch() == b"gfx950"


def _prototypes():
    """name -> (return type, [parameter C types]) for every `rac_*` prototype of include/rac_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "rac_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    out = {}
    for ret, name, params in re.findall(r"\b(int|int64_t|const char\*)\s+(rac_\w+)\s*\(([^)]*)\)\s*;", hdr):
        params = " ".join(params.split())
        types = []
        if params != "void":
            for prm in params.split(","):
                prm = prm.strip()
                m = re.match(r"^(.*?)(\w+)(\[\w*\])?$", prm)  # type, name, optional array suffix
                types.append((m.group(1).strip() + (" *" if m.group(3) else "")).replace(" *", "*").replace("* ", "*"))
        out[name] = (ret, types)
    return out


def test_ctypes_signatures_match_the_header():
    """_lib._SIGS is written by hand: every export's arity and every parameter's type class is checked against the
    prototype the header declares (a missing / extra / reordered argument would otherwise shift everything behind it)."""
    import ctypes as C
    from robot_aware_control_amd import _lib
    protos = _prototypes()
    assert set(protos) == set(_lib._SIGS)
    structs = {"rac_conv_args": _lib.ConvArgs, "rac_wgrad_args": _lib.WgradArgs, "rac_grad_src": _lib.GradSrc}

    def expected(ctype):
        t = re.sub(r"\bconst\b", "", ctype).replace(" ", "")
        if t in ("int32_t", "int"):
            return C.c_int32
        if t == "int64_t":
            return C.c_int64
        if t == "float":
            return C.c_float
        m = re.match(r"^(rac_\w+)\*$", t)
        if m and m.group(1) in structs:
            return C.POINTER(structs[m.group(1)])
        if t.endswith("**"):       # host array of device pointers
            return C.POINTER(C.c_void_p)
        assert t.endswith("*"), ctype
        return C.c_void_p

    for name, (ret, types) in protos.items():
        sig = _lib._SIGS[name]
        assert len(sig) == len(types), (name, len(sig), types)
        for i, (have, ctype) in enumerate(zip(sig, types)):
            assert have is expected(ctype) or have == expected(ctype), (name, i, ctype, have)
        want_ret = {"int": C.c_int, "int64_t": C.c_int64, "const char*": C.c_char_p}[ret]
        assert _lib._RET.get(name, C.c_int) is want_ret, (name, ret)


def test_compiled_consumer_sees_the_same_structs():
    """tests/abi_consumer.cpp (C++, include/rac_hip.h, -lrac_hip; built by __graft_entry__.build()): the library answers
    with the header's ABI version, and the struct layouts a C++ translation unit sees are the ctypes mirrors'."""
    import ctypes as C
    import subprocess
    import __graft_entry__ as entry
    from robot_aware_control_amd import _lib
    exe = entry.build_abi_consumer()
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    kv = dict(line.split() for line in res.stdout.splitlines())
    assert int(kv["rac_version"]) == int(kv["RAC_ABI_VERSION"]) == _lib.ABI_VERSION and kv["arch"] == "gfx950"
    for cname, cls in (("rac_conv_args", _lib.ConvArgs), ("rac_wgrad_args", _lib.WgradArgs),
                       ("rac_absmax_job", _lib.AbsmaxJob), ("rac_frag_job", _lib.FragJob),
                       ("rac_grad_src", _lib.GradSrc)):
        assert int(kv[f"sizeof_{cname}"]) == C.sizeof(cls), cname
    for key, val in kv.items():
        m = re.match(r"offsetof_(rac_conv_args|rac_wgrad_args|rac_frag_job|rac_grad_src)_(\w+)", key)
        if m:
            cls = {"rac_conv_args": _lib.ConvArgs, "rac_wgrad_args": _lib.WgradArgs, "rac_frag_job": _lib.FragJob,
                   "rac_grad_src": _lib.GradSrc}[m.group(1)]
            assert getattr(cls, m.group(2)).offset == int(val), key


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly off the GPU."""
    from robot_aware_control_amd import RacError, ops
    x = torch.zeros(1, 8, 8, 4)
    w = torch.zeros(4, 3, 3, 4).permute(0, 3, 1, 2)
    with pytest.raises(RacError):
        ops.conv_forward(x, None, w)
    from robot_aware_control_amd.image import zero_robot_region
    with pytest.raises(RacError):
        zero_robot_region(torch.zeros(1, 1, 4, 4), torch.zeros(1, 3, 4, 4))
    out = zero_robot_region(np.array([[True, False]]), np.array([[5, 7]]))  # numpy branch stays on the host
    assert out.tolist() == [[0, 7]]


def test_config_matches_reference_flags(monkeypatch):
    from robot_aware_control_amd import config
    # the robot-aware README command (reference README.md:111) parses, booleans are strings
    argv = ("--jobname ra --wandb False --data_root data --batch_size 16 --n_future 5 --n_past 1 --n_eval 6 "
            "--g_dim 512 --z_dim 64 --model svg --niter 1000 --epoch_size 300 --eval_interval 15 "
            "--checkpoint_interval 5 --reconstruction_loss dontcare_l1 --last_frame_skip True "
            "--scheduled_sampling True --action_dim 5 --robot_dim 5 --data_threads 5 --lr 0.0001 "
            "--experiment train_robonet --preprocess_action raw --world_error_dict x.pkl --train_val_split 0.95 "
            "--model_use_robot_state True --model_use_mask True --model_use_future_mask True "
            "--random_snippet True --image_width 64 --image_height 64").split()
    cf, unparsed = config.argparser(argv)
    assert unparsed == [] and cf.g_dim == 512 and cf.model_use_mask is True and cf.last_frame_skip is True
    assert cf.reconstruction_loss == "dontcare_l1" and cf.lr == 1e-4
    # reference defaults (src/config/__init__.py:151-249,315-357)
    d, _ = config.argparser([])
    assert (d.image_height, d.image_width, d.z_dim, d.g_dim, d.n_future, d.beta, d.beta1) == (48, 64, 10, 128, 9, 1e-4, 0.9)
    assert (d.candidates_batch_size, d.horizon, d.opt_iter, d.topk, d.action_candidates) == (200, 5, 10, 5, 30)
    assert d.robot_dim == 6 and d.model_use_robot_state is True and d.scheduled_sampling_k == 4000
    # stale-but-published flags are accepted; unknown ones abort like the reference
    config.argparser(["--stoch", "True", "--experiment", "singlerobot", "--multiview", "False"])
    with pytest.raises(AssertionError):
        config.argparser(["--definitely_not_a_flag", "1"])
    with pytest.raises(SystemExit):
        config.argparser(["--reconstruction_loss", "l7"])


def test_shard_bounds_cover_and_balance():
    from robot_aware_control_amd.trajectory_sampler import shard_bounds
    for n in (1, 7, 8, 1000, 1001, 8000, 8001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _ns(**kw):
    d = dict(device=torch.device("cpu"), image_width=64, image_height=64, channels=3, model_use_mask=True,
             model_use_future_mask=True, model_use_heatmap=False, model_use_future_heatmap=False,
             model_use_robot_state=True, model_use_future_robot_state=False, g_dim=64, z_dim=16, action_dim=5,
             robot_dim=5, batch_size=2, lstm_group_norm=False, last_frame_skip=True)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.mark.parametrize("flags", [dict(), dict(model_use_mask=False, model_use_future_mask=False,
                                                model_use_robot_state=False),
                                   dict(model_use_future_robot_state=True, g_dim=32, z_dim=10)])
def test_state_dict_matches_reference_inventory(flags):
    from robot_aware_control_amd.model import SVGConvModel
    ns = _ns(**flags)
    m = SVGConvModel(ns)
    spec = orc.param_spec(orc.cfg_from_namespace(ns))
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec]
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == tuple(shape), k
    # conv weights are stored [Cout][k][k][Cin]; all parameters alias one flat buffer
    w = m.encoder.c2[0].main[0].weight
    assert w.stride() == (9 * 64, 1, 3 * 64, 64)
    flat, grad = m.flat_parameters()
    assert w.data_ptr() >= flat.data_ptr() and w.data_ptr() < flat.data_ptr() + 4 * flat.numel()
    # load_state_dict writes through the views
    ws = orc.make_weights(orc.cfg_from_namespace(ns), seed=3)
    m.load_state_dict(ws)
    assert torch.equal(m.state_dict()["prior.lstm.0.gates.weight"], ws["prior.lstm.0.gates.weight"])
    assert w.data_ptr() >= flat.data_ptr()  # still aliased
    with pytest.raises(ValueError):
        SVGConvModel(_ns(image_width=48))
    # NormConvLSTMCell variant (--lstm_group_norm True): reference key names
    ns_gn = _ns(lstm_group_norm=True)
    sd_gn = SVGConvModel(ns_gn).state_dict()
    spec_gn = orc.param_spec(orc.cfg_from_namespace(ns_gn))
    assert list(sd_gn.keys()) == [k for k, _, _ in spec_gn]
    assert "prior.lstm.0.ih_gates.1.weight" in sd_gn and "frame_predictor.lstm.1.c_norm.bias" in sd_gn


def test_fused_adam_state_dict_is_torch_adam_compatible():
    from robot_aware_control_amd.model import SVGConvModel
    from robot_aware_control_amd.optim import FusedAdam
    m = SVGConvModel(_ns(g_dim=32, z_dim=8))
    opt = FusedAdam(m, lr=1e-4, betas=(0.9, 0.999))
    opt._moments()[0].fill_(0.5)
    opt._steps = 3
    sd = opt.state_dict()
    n_params = len(list(m.parameters()))
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == n_params
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m.parameters()], lr=1e-4)
    ref.load_state_dict(sd)  # the reference trainer's optimizer accepts it (trainer.py:877)
    opt2 = FusedAdam(m, lr=1e-4)
    opt2.load_state_dict(sd)
    assert opt2._steps == 3 and float(opt2._moments()[0][0]) == 0.5

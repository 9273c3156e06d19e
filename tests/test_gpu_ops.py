"""GPU parity of every HIP kernel against PyTorch-CPU fp32 / the oracle, through the C ABI.

Tolerances: conv-type kernels 2e-5 relative to the output scale (fp32 MFMA, different
summation order than ATen-CPU); elementwise kernels 1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import svg_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.Philox(key=[seed, 77]))
    return torch.from_numpy(g.standard_normal(shape, dtype=np.float32) * np.float32(scale))


def cl_weight(w):
    """logical (Cout,Cin,k,k) tensor stored [Cout][k][k][Cin]."""
    return w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def to_map(x, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(dev)


def from_map(m):
    return m.permute(0, 3, 1, 2).cpu()


def relerr(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CONV_CASES = [
    # B, H, W, C0, C1, Cout, k
    (2, 8, 8, 64, 0, 64, 3),
    (2, 8, 8, 128, 128, 512, 5),     # ConvLSTM layer 0 shape (g=128)
    (3, 8, 8, 64, 64, 256, 3),       # ConvLSTM layer 1, odd batch (M tail)
    (2, 64, 64, 5, 0, 64, 3),        # first encoder layer, unaligned Cin (scalar gather path)
    (2, 8, 8, 74, 0, 64, 3),         # prior_input_conv, Cin % 4 != 0
    (2, 8, 8, 64, 0, 16, 3),         # mu/logvar head, narrow-N tile
    (1, 6, 8, 32, 0, 32, 3),         # 48x64 frame -> 6x8 latent
    (4, 16, 16, 256, 256, 256, 3),   # decoder upc3.0 virtual concat
    (2, 32, 32, 128, 0, 128, 3),
    (20, 8, 8, 256, 0, 512, 3),      # large-M 128x128 tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    from robot_aware_control_amd import ops
    B, H, W, C0, C1, Cout, k = case
    Cin = C0 + C1
    x = rnd(1, B, Cin, H, W).requires_grad_(True)
    w = (rnd(2, Cout, Cin, k, k) * (1.0 / np.sqrt(Cin * k * k))).requires_grad_(True)
    b = rnd(3, Cout, scale=0.1).requires_grad_(True)
    y_ref = F.conv2d(x, w, b, 1, k // 2)
    gy = rnd(4, *y_ref.shape)
    y_ref.backward(gy)

    x0 = to_map(x.detach()[:, :C0], dev).requires_grad_(True)
    x1 = to_map(x.detach()[:, C0:], dev).requires_grad_(True) if C1 else None
    wd = cl_weight(w.detach()).to(dev).requires_grad_(True)
    assert wd.stride() == (k * k * Cin, 1, k * Cin, Cin)
    bd = b.detach().to(dev).requires_grad_(True)
    y = ops.ConvBias.apply(x0, x1, wd, bd, ops.ACT_NONE)
    assert relerr(from_map(y), y_ref.detach()) < 2e-5
    y.backward(to_map(gy, dev))
    gx = torch.cat([from_map(x0.grad)] + ([from_map(x1.grad)] if C1 else []), 1)
    assert relerr(gx, x.grad) < 2e-5
    assert relerr(wd.grad.cpu(), w.grad) < 3e-5
    assert relerr(bd.grad.cpu(), b.grad) < 2e-5
    # accumulate semantics: a second backward doubles the parameter gradients
    y2 = ops.ConvBias.apply(x0, x1, wd, bd, ops.ACT_NONE)
    y2.backward(to_map(gy, dev))
    assert relerr(wd.grad.cpu(), 2 * w.grad) < 3e-5
    assert relerr(bd.grad.cpu(), 2 * b.grad) < 2e-5


def test_conv_mfma_layout_asymmetric(dev):
    """Integer data, asymmetric weights: any row/col swap of the MFMA C/D map shows up exactly."""
    from robot_aware_control_amd import ops
    B, H, W, Cin, Cout, k = 1, 8, 8, 32, 64, 3
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    x = torch.from_numpy(g.integers(-3, 4, (B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy(g.integers(-3, 4, (Cout, Cin, k, k)).astype(np.float32))
    y_ref = F.conv2d(x, w, None, 1, 1)
    y = ops.conv_forward(to_map(x, dev), None, cl_weight(w).to(dev))
    assert torch.equal(from_map(y), y_ref)


@pytest.mark.parametrize("B,H,W,Ci", [(2, 64, 64, 64), (1, 48, 64, 64), (1, 128, 128, 32), (2, 32, 32, 64)])
def test_convT_head(dev, B, H, W, Ci, monkeypatch):
    """ConvTranspose2d + bias + Sigmoid head.  64 -> 4 channels on H % 8 = W % 16 = 0 maps: the matrix pipe with the
    roles swapped (rac_head_fwd_split); RAC_HEAD_MFMA=0 or the exact-fp32 mode: the FMA kernel (rac_head_fwd, weights as
    scalar operands); otherwise the narrow 32-column form of the rows kernel (4 real columns)."""
    from robot_aware_control_amd import ops
    x = rnd(1, B, Ci, H, W).requires_grad_(True)
    w = (rnd(2, Ci, 4, 3, 3) * 0.05).requires_grad_(True)
    b = rnd(3, 4, scale=0.1).requires_grad_(True)
    y_ref = torch.sigmoid(F.conv_transpose2d(x, w, b, 1, 1))
    gy = rnd(4, *y_ref.shape)
    y_ref.backward(gy)
    xd = to_map(x.detach(), dev).requires_grad_(True)
    wd = cl_weight(w.detach()).to(dev).requires_grad_(True)
    bd = b.detach().to(dev).requires_grad_(True)
    y = ops.ConvTHead.apply(xd, wd, bd)
    assert relerr(from_map(y), y_ref.detach()) < 1e-5
    if Ci == 64:  # the three forms against fp64
        y64 = torch.sigmoid(F.conv_transpose2d(x.detach().double(), w.detach().double(), b.detach().double(), 1, 1))
        assert relerr(from_map(y), y64) < 2e-6
        monkeypatch.setattr(ops, "HEAD_MFMA", False)
        y_fma = ops.ConvTHead.apply(xd.detach(), wd.detach(), bd.detach())
        monkeypatch.setattr(ops, "HEAD_DIRECT", False)
        y_rows = ops.ConvTHead.apply(xd.detach(), wd.detach(), bd.detach())
        monkeypatch.setattr(ops, "HEAD_DIRECT", True)
        monkeypatch.setattr(ops, "HEAD_MFMA", True)
        assert relerr(from_map(y_fma), y64) < 2e-6 and relerr(from_map(y_rows), y64) < 2e-6
        assert relerr(from_map(y.detach()), from_map(y_fma).double()) < 2e-6
        # the frozen model's form (one scale per image): an image's output is the same bits alone and in a batch
        with torch.no_grad():
            yf = ops.ConvTHead.apply(xd.detach(), wd.detach(), bd.detach(), True)
            y1 = ops.ConvTHead.apply(xd.detach()[B - 1:].contiguous(), wd.detach(), bd.detach(), True)
        assert torch.equal(y1, yf[B - 1:]) and relerr(from_map(yf), y64) < 2e-6
    y.backward(to_map(gy, dev))
    assert relerr(from_map(xd.grad), x.grad) < 2e-5
    assert relerr(wd.grad.cpu(), w.grad) < 3e-5
    assert relerr(bd.grad.cpu(), b.grad) < 2e-5
    if Ci == 64 and W % 16 == 0:
        # the head's data gradient as one streaming pass (rac_head_dgrad) against the exact-fp32 implicit GEMM and fp64
        x64 = x.detach().double().requires_grad_(True)
        torch.sigmoid(F.conv_transpose2d(x64, w.detach().double(), b.detach().double(), 1, 1)).backward(gy.double())
        monkeypatch.setattr(ops, "HEAD_DGRAD", False)
        xg = to_map(x.detach(), dev).requires_grad_(True)
        ops.ConvTHead.apply(xg, wd.detach(), bd.detach()).backward(to_map(gy, dev))
        monkeypatch.setattr(ops, "HEAD_DGRAD", True)
        e_new, e_gemm = relerr(from_map(xd.grad), x64.grad), relerr(from_map(xg.grad), x64.grad)
        assert e_new < 2e-6 and e_new < 2 * e_gemm + 1e-7, (e_new, e_gemm)


def test_first_layer_and_head_on_constant_inputs(dev):
    """Edge cases of the two matrix-pipe frame kernels' self-made operand scales: an all-zero frame (every pixel's own
    maximum is 0: scale 1, output = act(shift)), a frame of ones next to it in the batch, and an all-zero head input
    (output = sigmoid(bias)); one huge pixel does not cost its neighbours precision (per-pixel scales)."""
    from robot_aware_control_amd import ops
    w = rnd(5, 64, 3, 3, 3) * 0.2
    scale, shift = rnd(6, 64).abs() + 0.5, rnd(7, 64, scale=0.3)
    img = torch.zeros(3, 3, 32, 32)
    img[1] = 1.0
    img[2] = torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(1)) * 1e-3
    img[2, :, 7, 9] = 5e4                                     # 5e7 times its neighbours
    ref = F.leaky_relu(F.conv2d(img.double(), w.double(), None, 1, 1) * scale.double().view(1, -1, 1, 1)
                       + shift.double().view(1, -1, 1, 1), 0.2)
    out = from_map(ops.first_layer_frozen(img.to(dev), None, None, cl_weight(w).to(dev), scale.to(dev), shift.to(dev)))
    assert torch.equal(out[0], F.leaky_relu(shift.view(-1, 1, 1).expand(64, 32, 32), 0.2))
    assert relerr(out[1], ref[1]) < 2e-6
    far = torch.ones(32, 32, dtype=torch.bool)
    far[5:10, 7:12] = False                                   # pixels whose 3 x 3 window does not see the huge one
    assert float(((out[2].double() - ref[2]).abs()[:, far] / ref[2].abs()[:, far].clamp_min(1e-3)).max()) < 2e-6
    assert relerr(out[2], ref[2]) < 2e-6
    wh = (rnd(2, 64, 4, 3, 3) * 0.05)
    b = rnd(3, 4, scale=0.1)
    x = torch.zeros(2, 16, 16, 64, device=dev)
    with torch.no_grad():
        y = ops.ConvTHead.apply(x, cl_weight(wh).to(dev), b.to(dev), True)
    assert torch.allclose(y.cpu(), torch.sigmoid(b).view(1, 1, 1, 4).expand(2, 16, 16, 4), atol=1e-7)


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 0, 128), (16, 8, 8, 256, 0, 512), (2, 32, 32, 128, 128, 64),
                                   (2, 64, 64, 5, 0, 64)])
def test_vgg_layer_train_and_eval(dev, shape):
    from robot_aware_control_amd import ops
    B, H, W, C0, C1, Cout = shape
    Cin = C0 + C1
    x = rnd(1, B, Cin, H, W).requires_grad_(True)
    w = (rnd(2, Cout, Cin, 3, 3) * (1.4 / np.sqrt(Cin * 9))).requires_grad_(True)
    gamma = (1 + rnd(3, Cout, scale=0.1)).requires_grad_(True)
    beta = rnd(4, Cout, scale=0.1).requires_grad_(True)
    rm, rv = rnd(5, Cout, scale=0.1), 1 + rnd(6, Cout, scale=0.1).abs()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    x0 = to_map(x.detach()[:, :C0], dev).requires_grad_(True)
    x1 = to_map(x.detach()[:, C0:], dev).requires_grad_(True) if C1 else None
    wd = cl_weight(w.detach()).to(dev).requires_grad_(True)
    gd, bd = gamma.detach().to(dev).requires_grad_(True), beta.detach().to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    y = ops.VggLayer.apply(x0, x1, wd, gd, bd, rmd, rvd, True, 2, None)
    # The reference uses the GPU's LeakyReLU slope pattern: a pre-activation within rounding distance of zero
    # may take the other slope, which changes the backward pass discontinuously (see test_gpu_model.py).
    z_ref = F.batch_norm(F.conv2d(x, w, None, 1, 1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    mask = from_map(y) > 0
    flips = (z_ref.detach() > 0) != mask
    assert int(flips.sum()) <= 3 and float(z_ref.detach()[flips].abs().max() if flips.any() else 0.0) < 1e-5
    y_ref = z_ref * torch.where(mask, 1.0, 0.2)
    gy = rnd(7, *y_ref.shape)
    y_ref.backward(gy)
    # second momentum update on the same batch (the double encoder pass of the reference)
    with torch.no_grad():
        F.batch_norm(F.conv2d(x, w, None, 1, 1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)

    assert relerr(from_map(y), y_ref.detach()) < 3e-5
    assert relerr(rmd.cpu(), rm_ref) < 1e-5 and relerr(rvd.cpu(), rv_ref) < 1e-5
    y.backward(to_map(gy, dev))
    gx = torch.cat([from_map(x0.grad)] + ([from_map(x1.grad)] if C1 else []), 1)
    assert relerr(gx, x.grad) < 1e-4
    assert relerr(wd.grad.cpu(), w.grad) < 1e-4
    assert relerr(gd.grad.cpu(), gamma.grad) < 1e-4 and relerr(bd.grad.cpu(), beta.grad) < 1e-4
    # eval: folded BatchNorm in the conv epilogue
    y_eval_ref = F.leaky_relu(F.batch_norm(F.conv2d(x, w, None, 1, 1), rm, rv, gamma, beta, False, 0.1, 1e-5), 0.2)
    scale = gamma.detach() / torch.sqrt(rv + 1e-5)
    shift = beta.detach() - rm * scale
    with torch.no_grad():
        y_eval = ops.VggLayer.apply(x0, x1, wd, gd, bd, rm.to(dev), rv.to(dev), False, 1, (scale.to(dev), shift.to(dev)))
    assert relerr(from_map(y_eval), y_eval_ref.detach()) < 2e-5


@pytest.mark.parametrize("g,k,B", [(64, 5, 2), (128, 3, 3), (512, 5, 2)])
def test_lstm_cell(dev, g, k, B):
    from robot_aware_control_amd import ops
    H = W = 8
    x, h0, c0 = [rnd(i, B, g, H, W, scale=0.7).requires_grad_(True) for i in (1, 2, 3)]
    w = (rnd(4, 4 * g, 2 * g, k, k) * (1.0 / np.sqrt(2 * g * k * k))).requires_grad_(True)
    b = rnd(5, 4 * g, scale=0.1).requires_grad_(True)
    sd = {"p.lstm.0.gates.weight": w, "p.lstm.0.gates.bias": b, "p.lstm.1.gates.weight": w, "p.lstm.1.gates.bias": b}
    h_ref, c_ref = orc.convlstm_cell(sd, "p", 0 if k == 5 else 1, x, (h0, c0))
    gh, gc = rnd(6, *h_ref.shape), rnd(7, *c_ref.shape)
    (h_ref * gh).sum().add((c_ref * gc).sum()).backward()
    xd, hd, cd = [to_map(t.detach(), dev).requires_grad_(True) for t in (x, h0, c0)]
    wd = cl_weight(w.detach()).to(dev).requires_grad_(True)
    bd = b.detach().to(dev).requires_grad_(True)
    h, c = ops.LstmCell.apply(xd, hd, cd, wd, bd)
    assert relerr(from_map(h), h_ref.detach()) < 2e-5 and relerr(from_map(c), c_ref.detach()) < 2e-5
    torch.autograd.backward([h, c], [to_map(gh, dev), to_map(gc, dev)])
    for got, ref in ((xd, x), (hd, h0), (cd, c0)):
        assert relerr(from_map(got.grad), ref.grad) < 5e-5
    assert relerr(wd.grad.cpu(), w.grad) < 5e-5
    assert relerr(bd.grad.cpu(), b.grad) < 5e-5


@pytest.mark.parametrize("mfma", [True, False], ids=["matrix_pipe", "fma"])
@pytest.mark.parametrize("B,H,W,Cm,zero", [(3, 64, 64, 2, True), (2, 48, 64, 1, True), (2, 32, 16, 0, False), (1, 16, 16, 4, True),
                                           (2, 32, 32, 5, False)])
def test_first_layer_from_planes(dev, B, H, W, Cm, zero, mfma, monkeypatch):
    """Frozen model's first encoder layer straight from the NCHW planes: zero_robot_region + mask concat + conv3x3 +
    folded BatchNorm + LeakyReLU(0.2) in one kernel, and the max |out| it leaves for the next conv -- on the matrix
    pipe (rac_first_layer_fwd_split: every pixel / channel scaled by its own maximum) and as exact-fp32 FMAs."""
    from robot_aware_control_amd import ops
    monkeypatch.setattr(ops, "FIRST_MFMA", mfma)
    img = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
    mask = (rnd(2, B, Cm, H, W) > 0.3).float() if Cm else None
    zm = (rnd(4, B, 1, H, W) > 0.5).float() if zero else None
    w = rnd(5, 64, 3 + Cm, 3, 3) * 0.2
    scale, shift = rnd(6, 64).abs() + 0.5, rnd(7, 64, scale=0.3)
    x = img * (1 - zm) if zero else img
    x = torch.cat([x, mask], 1) if Cm else x
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None, 1, 1) * scale.double().view(1, -1, 1, 1)
                       + shift.double().view(1, -1, 1, 1), 0.2)
    wd = cl_weight(w).to(dev)
    assert ops.first_layer_ok(img.to(dev), None if mask is None else mask.to(dev), wd)
    out = ops.first_layer_frozen(img.to(dev), None if zm is None else zm.to(dev), None if mask is None else mask.to(dev),
                                 wd, scale.to(dev), shift.to(dev))
    assert relerr(from_map(out), ref) < 2e-6
    # one maximum per image (the frozen model scales every image on its own)
    assert torch.equal(ops.amax_tag(out).cpu(), out.abs().amax((1, 2, 3)).view(torch.int32).cpu())
    # an image's result is the same bits alone and in the batch
    one = ops.first_layer_frozen(img[B - 1:].to(dev), None if zm is None else zm[B - 1:].to(dev),
                                 None if mask is None else mask[B - 1:].to(dev), wd, scale.to(dev), shift.to(dev))
    assert torch.equal(one, out[B - 1:])


@pytest.mark.parametrize("B,h,w,C0,C1,Cout", [(2, 8, 8, 64, 64, 64), (3, 32, 32, 64, 64, 64), (2, 16, 16, 128, 128, 128),
                                               (1, 24, 32, 32, 32, 64)])
def test_conv_reads_half_resolution_source(dev, B, h, w, C0, C1, Cout):
    """Frozen decoder: conv over [UpsamplingNearest2d(2)(d) | skip] with d read in place at (y / 2, x / 2) -- bit-equal
    to the conv over the materialised upsampled tensor."""
    from robot_aware_control_amd import ops
    d = to_map(rnd(31, B, C0, h, w), dev)
    sk = to_map(rnd(32, B, C1, 2 * h, 2 * w), dev)
    wt = cl_weight(rnd(33, Cout, C0 + C1, 3, 3) * 0.05).to(dev)
    scale, shift = (rnd(34, Cout).abs() + 0.5).to(dev), rnd(35, Cout, scale=0.2).to(dev)
    assert ops.vgg_up_frozen_ok(d, sk, wt)
    got = ops.vgg_up_frozen(d, sk, wt, scale, shift)
    up = ops.Upsample2.apply(d)
    want = ops.conv_forward_split(up, sk, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True)
    assert torch.equal(got, want)
    ref = F.leaky_relu(F.conv2d(torch.cat([F.interpolate(from_map(d).double(), scale_factor=2, mode="nearest"),
                                           from_map(sk).double()], 1), wt.cpu().double(), None, 1, 1)
                       * scale.cpu().double().view(1, -1, 1, 1) + shift.cpu().double().view(1, -1, 1, 1), 0.2)
    assert relerr(from_map(got), ref) < 2e-6


@pytest.mark.parametrize("B,H,W,ct,stride", [(2, 32, 48, 4, 4), (3, 64, 64, 5, 32), (1, 16, 16, 7, 8), (600, 16, 16, 3, 4)])
def test_thin_weight_gradient(dev, B, H, W, ct, stride):
    """dW of a 3x3 conv between a 64-channel map and a thin one (first encoder layer, output head) against fp64;
    accumulates into .grad; bit-reproducible (per-workgroup partials, fixed-order sum)."""
    from robot_aware_control_amd import ops
    wide = rnd(41, B, 64, H, W) * 1e-3
    thin = rnd(42, B, ct, H, W)
    wr = torch.zeros(64, ct, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(thin.double(), wr, None, 1, 1).backward(wide.double())
    thin_map = torch.zeros(B, H, W, stride)
    thin_map[..., :ct] = thin.permute(0, 2, 3, 1)
    wd = cl_weight(torch.zeros(64, ct, 3, 3)).to(dev).requires_grad_(True)
    ops.thin_wgrad_acc(to_map(wide, dev), thin_map.to(dev), ct, wd)
    first = wd.grad.clone()
    assert relerr(first.cpu(), wr.grad) < 3e-6
    ops.thin_wgrad_acc(to_map(wide, dev), thin_map.to(dev), ct, wd)
    assert relerr(wd.grad.cpu(), 2 * wr.grad) < 3e-6
    wd2 = cl_weight(torch.zeros(64, ct, 3, 3)).to(dev).requires_grad_(True)
    ops.thin_wgrad_acc(to_map(wide, dev), thin_map.to(dev), ct, wd2)
    assert torch.equal(wd2.grad, first)


def test_pool_upsample_tilecat(dev):
    from robot_aware_control_amd import ops
    x = rnd(1, 2, 12, 16, 16).requires_grad_(True)
    y_ref = F.max_pool2d(x, 2, 2)
    gy = rnd(2, *y_ref.shape)
    y_ref.backward(gy)
    xd = to_map(x.detach(), dev).requires_grad_(True)
    y = ops.MaxPool2.apply(xd)
    assert torch.equal(from_map(y), y_ref.detach())
    y.backward(to_map(gy, dev))
    assert torch.equal(from_map(xd.grad), x.grad)

    x = rnd(3, 2, 6, 4, 4).requires_grad_(True)  # C=6: scalar path
    y_ref = F.interpolate(x, scale_factor=2, mode="nearest")
    gy = rnd(4, *y_ref.shape)
    y_ref.backward(gy)
    xd = to_map(x.detach(), dev).requires_grad_(True)
    y = ops.Upsample2.apply(xd)
    assert torch.equal(from_map(y), y_ref.detach())
    y.backward(to_map(gy, dev))
    assert relerr(from_map(xd.grad), x.grad) < 1e-6

    B, g, z = 3, 8, 4
    a, r = rnd(5, B, 5), rnd(6, B, 5)
    hmap, zmap = rnd(7, B, g, 8, 8).requires_grad_(True), rnd(8, B, z, 8, 8).requires_grad_(True)
    ref = torch.cat([orc._tile(a, 8, 8), orc._tile(r, 8, 8), hmap, zmap], 1)
    gy = rnd(9, *ref.shape)
    ref.backward(gy)
    hd, zd = to_map(hmap.detach(), dev).requires_grad_(True), to_map(zmap.detach(), dev).requires_grad_(True)
    out = ops.TileCat.apply(a.to(dev), r.to(dev), None, hd, zd)  # 22 channels + 2 zero pad channels
    assert out.shape[3] == 24 and float(out[..., 22:].abs().max()) == 0.0
    assert torch.equal(from_map(out)[:, :22], ref.detach())
    out.backward(torch.cat([to_map(gy, dev), torch.zeros(B, 8, 8, 2, device=dev)], 3))
    assert torch.equal(from_map(hd.grad), hmap.grad) and torch.equal(from_map(zd.grad), zmap.grad)


def test_frame_ops_and_losses(dev, golden_dir):
    import os
    from robot_aware_control_amd import ops
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    target, mask, bw = (torch.from_numpy(g[k]) for k in ("target", "mask", "bw"))
    for name, kind, rw, use_bw in (("l1", "l1", 0, False), ("l1_bw", "l1", 0, True), ("mse", "mse", 0, False),
                                   ("dc_l1_w0", "dontcare_l1", 0.0, False), ("dc_l1_w05", "dontcare_l1", 0.5, False),
                                   ("dc_l1_w0_bw", "dontcare_l1", 0.0, True),
                                   ("dc_mse_w05", "dontcare_mse", 0.5, False)):
        pred = torch.from_numpy(g["pred"]).to(dev).requires_grad_(True)
        out = ops.ReconLoss.apply(pred, target.to(dev), mask.to(dev), bw.to(dev) if use_bw else None,
                                  ops.LOSS_KINDS[kind], rw)
        np.testing.assert_allclose(out[0].item(), g[name], rtol=2e-6)
        np.testing.assert_allclose(out[1].item(), g["robot_mse"], rtol=2e-6)
        np.testing.assert_allclose(out[2].item(), g["world_mse"], rtol=2e-6)
        out[0].backward()
        np.testing.assert_allclose(pred.grad.cpu().numpy(), g[name + "_grad"], rtol=1e-5, atol=1e-9)
    ts = [torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("mu1", "lv1", "mu2", "lv2")]
    kl = ops.KLLoss.apply(*ts, 3)
    np.testing.assert_allclose(kl.item(), g["kl"], rtol=3e-6)
    kl.backward()
    for t, k in zip(ts, ("kl_gmu1", "kl_glv1", "kl_gmu2", "kl_glv2")):
        np.testing.assert_allclose(t.grad.cpu().numpy(), g[k], rtol=2e-5, atol=1e-7)

    # composite / zero-region / pack-input against the oracle expressions
    x4 = torch.sigmoid(rnd(1, 2, 4, 16, 16)).requires_grad_(True)
    prev = rnd(2, 2, 3, 16, 16).abs().requires_grad_(True)
    m = (rnd(3, 2, 1, 16, 16) > 0.5).float()
    ref = orc.zero_robot_region(m, orc.composite(x4, prev))
    gy = rnd(4, *ref.shape)
    ref.backward(gy)
    x4d = to_map(x4.detach(), dev).requires_grad_(True)
    pd = prev.detach().to(dev).requires_grad_(True)
    out = ops.ZeroRegion.apply(ops.Composite.apply(x4d, pd), m.to(dev))
    assert relerr(out.cpu(), ref.detach()) < 1e-6
    out.backward(gy.to(dev))
    assert relerr(from_map(x4d.grad), x4.grad) < 1e-6 and relerr(pd.grad.cpu(), prev.grad) < 1e-6
    img = rnd(5, 2, 3, 16, 16).requires_grad_(True)
    m2 = torch.cat([m, (rnd(6, 2, 1, 16, 16) > 0).float()], 1)
    ref = torch.cat([orc.zero_robot_region(m, img), m2], 1)
    gy = rnd(7, *ref.shape)
    ref.backward(gy)
    imgd = img.detach().to(dev).requires_grad_(True)
    out = ops.PackInput.apply(imgd, m.to(dev), m2.to(dev))  # 5 channels + 3 zero pad channels
    assert out.shape[3] == 8 and float(out[..., 5:].abs().max()) == 0.0
    assert torch.equal(from_map(out)[:, :5], ref.detach())
    out.backward(torch.cat([to_map(gy, dev), torch.zeros(2, 16, 16, 3, device=dev)], 3))
    assert torch.equal(imgd.grad.cpu(), img.grad)

    mu, lv, eps = [rnd(i, 2, 8, 8, 4).to(dev).requires_grad_(i < 12) for i in (10, 11, 12)]
    z = ops.Reparam.apply(mu, lv, eps)
    zr = eps.cpu() * torch.exp(0.5 * lv.detach().cpu()) + mu.detach().cpu()
    assert relerr(z.cpu(), zr) < 1e-6


def test_adam_matches_torch(dev):
    from robot_aware_control_amd import _lib
    n = 1003
    p, g = rnd(1, n), rnd(2, n, scale=1e-3)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999))
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        gg = g * step
        ref.grad = gg.clone()
        opt.step()
        gd = gg.to(dev)
        _lib.call("rac_adam_step", pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9,
                  0.999, 1e-8, step, _lib.stream_ptr())
        torch.cuda.synchronize()
        assert relerr(pd.cpu(), ref.detach()) < 1e-6


def test_cem_step_tail(dev, golden_dir):
    import os
    from robot_aware_control_amd import _lib
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    curr, goal = torch.from_numpy(g["c_curr"]), torch.from_numpy(g["c_goal"])
    N, _, H, W = curr.shape
    # identity compositing (m = 0) so that next == curr: isolates the cost reduction
    x4 = torch.zeros(N, H, W, 4, device=dev)
    curr_d, goal_d = curr.to(dev), goal.to(dev)  # keep the device copies alive across the raw-pointer call
    for kind, key, cm, gm in ((0, "cost_l2", None, None), (1, "cost_dontcare", g["c_cmask"], g["c_gmask"])):
        nxt = torch.empty(N, 3, H, W, device=dev)
        cost = torch.zeros(N, device=dev, dtype=torch.float64)
        cmd = torch.from_numpy(cm.astype(np.float32)).to(dev) if cm is not None else None
        gmd = torch.from_numpy(gm.astype(np.uint8)).to(dev) if gm is not None else None
        _lib.call("rac_cem_step_tail", x4.data_ptr(), curr_d.data_ptr(), None, goal_d.data_ptr(),
                  _lib.ptr(cmd), _lib.ptr(gmd), kind, 1.0, 1, nxt.data_ptr(), cost.data_ptr(), N, H * W,
                  _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(nxt.cpu(), curr)
        np.testing.assert_allclose(cost.cpu().numpy(), g[key].astype(np.float64), rtol=2e-6)


def frag_order16(w: torch.Tensor) -> torch.Tensor:
    """[Cout][k][k][Cin] memory -> contiguous [Cout/32][Cin/32][k*k][nb][q][co mod 16][8]: the operand registers of
    v_mfma_f32_16x16x32_f16, lane = 16 q + co mod 16, co = 32 tile + 16 nb + .., ci = 32 chunk + 8 q + j."""
    co, ci, k, _ = w.shape
    mem = w.permute(0, 2, 3, 1).reshape(co // 32, 2, 16, k * k, ci // 32, 4, 8)  # nt, nb, lr, tap, cc, q, j
    return mem.permute(0, 4, 3, 1, 5, 2, 6).contiguous()


def split_f16x2(x: torch.Tensor):
    """The two fp16 parts of x * 2^k with max |x| * 2^k in [2^14, 2^15) -- the operand format of csrc/rac_split16.hip."""
    amax = float(x.abs().max())
    k = 14 - int(np.floor(np.log2(amax))) if amax > 0 else 0
    v = x.double() * 2.0 ** k
    h1 = v.float().half()
    h2 = (v.float() - h1.float()).half()
    return h1, h2, k


SPLIT_CASES = [(2, 8, 8, 64, 64, 256, 5), (3, 8, 8, 128, 128, 512, 3), (20, 8, 8, 64, 0, 96, 3),
               (5, 4, 8, 32, 64, 160, 3), (4, 8, 8, 64, 64, 1024, 3), (5, 6, 8, 64, 64, 128, 5)]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_split_precision_f16x2(dev, case):
    """Two fp16 parts per scaled operand, three part-products, fp32 accumulation: as close to fp64 as the exact-fp32
    MFMA kernel, whatever the magnitude of the operands (the power-of-two scales come from device-side maxima)."""
    from robot_aware_control_amd import ops
    B, H, W, C0, C1, Cout, k = case
    Cin = C0 + C1
    assert ops.split_supported(H, W, k, Cin, Cout, C0 if C1 else 0) and H * W <= 128
    # exactness of the fragment layouts and of the scaling: small integers live entirely in the first part
    g = np.random.Generator(np.random.Philox(key=[9, 9]))
    xi = torch.from_numpy(g.integers(-3, 4, (B, Cin, H, W)).astype(np.float32))
    wi = torch.from_numpy(g.integers(-3, 4, (Cout, Cin, k, k)).astype(np.float32))
    x0, x1 = to_map(xi[:, :C0], dev), (to_map(xi[:, C0:], dev) if C1 else None)
    y = ops.conv_forward_split(x0, x1, cl_weight(wi).to(dev))
    assert torch.equal(from_map(y), F.conv2d(xi, wi, None, 1, k // 2))
    # accuracy on real-valued data against fp64, next to the exact-fp32 MFMA path; operands of very different
    # magnitude (gradients are ~1e-6, pre-activations ~1e2) and a heavy-tailed tensor (one outlier 1e4 x the rest)
    x = rnd(1, B, Cin, H, W)
    w = rnd(2, Cout, Cin, k, k) * (1.0 / np.sqrt(Cin * k * k))
    b = rnd(3, Cout, scale=0.1)
    for xs, ws, spike in ((1.0, 1.0, False), (3e-7, 40.0, False), (5e3, 1e-3, False), (1.0, 1.0, True)):
        xv, wv, bv = x * xs, w * ws, b * (xs * ws)
        if spike:
            xv = xv.clone()
            xv[0, 0, 0, 0] = 1e4
        ref = F.conv2d(xv.double(), wv.double(), bv.double(), 1, k // 2)
        x0, x1 = to_map(xv[:, :C0], dev), (to_map(xv[:, C0:], dev) if C1 else None)
        wd, bd = cl_weight(wv).to(dev), bv.to(dev)
        e_split = relerr(from_map(ops.conv_forward_split(x0, x1, wd, bd)), ref)
        e_fp32 = relerr(from_map(ops.conv_forward(x0, x1, wd, bd, allow_split=False)), ref)
        assert e_split < 2e-6 and e_split < 4 * e_fp32 + 2e-7, (xs, ws, spike, e_split, e_fp32)
    # data gradient = forward conv with the transposed, tap-flipped weight (same kernels)
    gy = rnd(4, B, Cout, H, W) * 1e-5
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, 1, k // 2).backward(gy.double())
    d0, d1 = ops.conv_dgrad_split(to_map(gy, dev), cl_weight(w).to(dev), C0, C1)
    got = torch.cat([from_map(d0)] + ([from_map(d1)] if C1 else []), 1)
    assert relerr(got, xr.grad) < 2e-6


@pytest.mark.parametrize("case", [(16, 8, 8, 128, 32), (300, 8, 8, 64, 64), (3, 16, 16, 64, 16)])
def test_gauss_head_is_one_conv(dev, case):
    """mu / logvar heads (lstm.py:273-274) as one conv over the two adjacent parameters: outputs, data gradient,
    weight and bias gradients (accumulated in place into the flat gradient views) against fp64."""
    from robot_aware_control_amd import ops
    B, H, W, g, z = case
    w = rnd(21, 2 * z, g, 3, 3) * 0.05
    b = rnd(22, 2 * z, scale=0.1)
    nw = 2 * z * g * 9
    flat, grad = torch.zeros(nw + 2 * z, device=dev), torch.zeros(nw + 2 * z, device=dev)
    shape, stride = (2 * z, g, 3, 3), (9 * g, 1, 3 * g, g)
    wm = torch.as_strided(flat, shape, stride, 0)
    wm.copy_(w)
    wm.requires_grad_(True)
    wm.grad = torch.as_strided(grad, shape, stride, 0)
    bm = flat[nw:]
    bm.copy_(b)
    bm.requires_grad_(True)
    bm.grad = grad[nw:]
    assert ops.gauss_head_ok((B, H, W, g), wm)
    x = rnd(23, B, g, H, W)
    gm, gl = rnd(24, B, z, H, W) * 1e-2, rnd(25, B, z, H, W) * 1e-3
    h = to_map(x, dev).requires_grad_(True)
    mu, lv = ops.GaussHead.apply(h, wm, bm)
    ((mu * to_map(gm, dev)).sum() + (lv * to_map(gl, dev)).sum()).backward()
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.conv2d(xr, wr, br, 1, 1)
    ((y[:, :z] * gm).sum() + (y[:, z:] * gl).sum()).backward()
    assert relerr(from_map(mu), y[:, :z]) < 2e-6 and relerr(from_map(lv), y[:, z:]) < 2e-6
    assert relerr(from_map(h.grad), xr.grad) < 2e-6
    assert relerr(wm.grad.cpu(), wr.grad) < 3e-6 and relerr(bm.grad.cpu(), br.grad) < 3e-6
    # only one of the two outputs carries gradient (sample_mean / a KL-only path)
    grad.zero_()
    h2 = to_map(x, dev).requires_grad_(True)
    mu, lv = ops.GaussHead.apply(h2, wm, bm)
    (mu * to_map(gm, dev)).sum().backward()
    xr.grad = None
    (F.conv2d(xr, w.double(), b.double(), 1, 1)[:, :z] * gm).sum().backward()
    assert relerr(from_map(h2.grad), xr.grad) < 2e-6 and float(wm.grad[z:].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(64, 96, 3), (128, 64, 5), (32, 64, 3)])
def test_weight_frag_split(dev, shape):
    """rac_absmax + rac_weight_frag_split == scale by 2^k, split into two fp16 parts, permute into MFMA fragment
    order (forward and data-gradient weight); the parts carry 22 bits of every weight."""
    from robot_aware_control_amd import ops
    co, ci, k = shape
    w = cl_weight(rnd(5, co, ci, k, k) * 0.02).to(dev)
    h1, h2, kexp = split_f16x2(w.cpu())
    parts, slot = ops.weight_parts(w)
    assert int(slot.cpu()) == int(w.abs().max().cpu().view(torch.int32))
    assert torch.equal(parts[0].cpu(), frag_order16(h1).flatten()) and torch.equal(parts[1].cpu(), frag_order16(h2).flatten())
    wt = w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous(memory_format=torch.channels_last)
    pt, _ = ops.weight_parts(w, transposed=True)
    t1, t2, _ = split_f16x2(wt.cpu())
    assert torch.equal(pt[0].cpu(), frag_order16(t1).flatten()) and torch.equal(pt[1].cpu(), frag_order16(t2).flatten())
    back = (h1.double() + h2.double()) * 2.0 ** -kexp
    assert float(((back - w.cpu().double()).abs() / w.cpu().double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -21
    # the maximum accumulates over calls and over two arrays
    a, b = rnd(6, 4096, scale=3.0).to(dev), rnd(7, 512, scale=5.0).to(dev)
    s2 = ops.amax_of(a, b)
    assert int(s2.cpu()) == int(torch.maximum(a.abs().max(), b.abs().max()).cpu().view(torch.int32))


@pytest.mark.parametrize("B,H,W,C0,C1,Cout,k", [(5, 8, 8, 64, 64, 256, 5), (3, 6, 8, 128, 0, 128, 3), (3, 16, 16, 64, 64, 128, 3),
                                                  (2, 64, 64, 64, 0, 64, 3), (4, 32, 32, 128, 0, 32, 3),
                                                  (80, 64, 64, 64, 0, 64, 3), (72, 64, 64, 32, 32, 32, 3)])
def test_per_image_scales_make_a_conv_batch_invariant(dev, B, H, W, C0, C1, Cout, k):
    """`per_image` (the frozen model): every image is scaled by its own max |x|, K is never split -> an image's output is
    the same bits alone, in any batch and at any position in it, even next to images 1e4 times larger; accuracy as the
    per-tensor form.  Producers leave one maximum per image (conv epilogue, tilecat, rac_absmax_rows)."""
    from robot_aware_control_amd import ops
    # (the two large batches are >= 2048 tiles of 128 pixels: the PERSISTENT rows kernel, whose sub-batches below take the
    # one-tile-per-workgroup kernel -- bit-equal)
    mags = torch.tensor([1.0, 3e-4, 2e3, 0.07, 11.0]).repeat((B + 4) // 5)[:B].view(B, 1, 1, 1)
    x0 = to_map(rnd(61, B, C0, H, W) * mags, dev)
    x1 = to_map(rnd(62, B, C1, H, W) * mags, dev) if C1 else None
    wt = cl_weight(rnd(63, Cout, C0 + C1, k, k) * 0.05).to(dev)
    bias = rnd(64, Cout, scale=0.1).to(dev)
    full = ops.conv_forward_split(x0, x1, wt, bias, per_image=True)
    slots = ops.amax_tag(full)
    assert slots.numel() == B
    assert torch.equal(slots.cpu(), full.abs().amax((1, 2, 3)).view(torch.int32).cpu())
    for order in ([B - 1], [1, 0], list(range(B - 1, -1, -1))):  # alone, a pair, everything reversed
        idx = torch.tensor(order, device=dev)
        sub = ops.conv_forward_split(x0[idx].contiguous(), None if x1 is None else x1[idx].contiguous(), wt, bias,
                                     per_image=True)
        assert torch.equal(sub, full[idx]), order
    x = from_map(x0) if x1 is None else torch.cat([from_map(x0), from_map(x1)], 1)
    ref = F.conv2d(x.double(), wt.cpu().double(), bias.cpu().double(), 1, k // 2)
    got = from_map(full).double()
    for b in range(B):  # every image to fp32 level against ITS OWN magnitude
        assert float((got[b] - ref[b]).abs().max() / ref[b].abs().max()) < 4e-6, b  # (K up to 3200 fp32 sums)
    # slabs for the ConvLSTM cell: one slab, same bits as the finished conv minus the bias
    slabs, n, _ = ops.conv_forward_split(x0, x1, wt, want_slabs=True, per_image=True)
    assert n == 1 and torch.equal(slabs[0], ops.conv_forward_split(x0, x1, wt, None, per_image=True))
    # measured per image when no producer left a tag
    plain = x0.clone()
    assert torch.equal(ops.amax_for(plain, per_image=True).cpu(), plain.abs().amax((1, 2, 3)).view(torch.int32).cpu())
    h = ops.tag_amax(x0.clone(), ops.amax_one(dev))  # a bound of 1 serves either granularity
    assert ops.amax_for(h, per_image=True).numel() == B and ops.amax_for(h).numel() == 1


def test_weight_parts_refreshed_in_one_launch(dev):
    """After an optimiser step (ops.PARAM_EPOCH) every registered weight is refreshed by one rac_absmax_multi + one
    rac_weight_frag_split_multi launch: same parts and maxima as the single-tensor calls."""
    import gc
    from robot_aware_control_amd import ops
    gc.collect()  # (models of earlier tests that only a reference cycle keeps alive: their weights are registered too)
    shapes = [(64, 96, 3, True), (128, 64, 5, False), (32, 64, 3, True), (256, 512, 5, True), (96, 32, 3, False)]
    ws = [cl_weight(rnd(30 + i, co, ci, k, k) * 0.02).to(dev) for i, (co, ci, k, _) in enumerate(shapes)]
    for w, (_, _, _, both) in zip(ws, shapes):  # registration: the single-tensor path
        ops.weight_parts(w)
        if both:
            ops.weight_parts(w, transposed=True)
    for i, w in enumerate(ws):  # "optimiser step": new values behind torch's version counters
        w.copy_(cl_weight(rnd(40 + i, *w.shape) * (0.5 + i)).to(dev))
    ops.PARAM_EPOCH += 1
    calls = []
    real = ops.call
    ops.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        got = [(ops.weight_parts(w), ops.weight_parts(w, transposed=True) if both else None)
               for w, (_, _, _, both) in zip(ws, shapes)]
    finally:
        ops.call = real
    # (zero-padded copies of other live models' weights, if any, are rebuilt first: not what this test counts)
    assert [c for c in calls if c != "rac_pad_rows"] == ["rac_absmax_multi", "rac_weight_frag_split_multi"], calls
    for w, ((parts, slot), tr) in zip(ws, got):
        h1, h2, _ = split_f16x2(w.cpu())
        assert int(slot.cpu()) == int(w.abs().max().cpu().view(torch.int32))
        assert torch.equal(parts[0].cpu(), frag_order16(h1).flatten()) and torch.equal(parts[1].cpu(), frag_order16(h2).flatten())
        if tr is not None:
            wt = w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous(memory_format=torch.channels_last)
            t1, t2, _ = split_f16x2(wt.cpu())
            assert torch.equal(tr[0][0].cpu(), frag_order16(t1).flatten()) and torch.equal(tr[0][1].cpu(), frag_order16(t2).flatten())


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 96, 3), (1, 32, 32, 64, 0, 128, 3), (3, 16, 16, 32, 0, 64, 5),
                                  (1, 16, 32, 32, 32, 160, 3), (2, 4, 64, 32, 0, 64, 3), (1, 64, 64, 64, 64, 64, 3),
                                  (2, 12, 16, 64, 0, 96, 3), (1, 24, 32, 32, 32, 64, 3)])
def test_conv_split_rows_kernel(dev, case):
    """The split-precision kernel for maps larger than a tile (whole image rows per tile + halo) against fp64, forward
    and data gradient, and exact on small integers."""
    from robot_aware_control_amd import ops
    B, H, W, C0, C1, Cout, k = case
    Cin = C0 + C1
    assert ops.split_supported(H, W, k, Cin, Cout, C0 if C1 else 0) and H * W > 128
    g = np.random.Generator(np.random.Philox(key=[11, 9]))
    xi = torch.from_numpy(g.integers(-3, 4, (B, Cin, H, W)).astype(np.float32))
    wi = torch.from_numpy(g.integers(-3, 4, (Cout, Cin, k, k)).astype(np.float32))
    x0, x1 = to_map(xi[:, :C0], dev), (to_map(xi[:, C0:], dev) if C1 else None)
    y = ops.conv_forward_split(x0, x1, cl_weight(wi).to(dev))
    assert torch.equal(from_map(y), F.conv2d(xi, wi, None, 1, k // 2))
    x = rnd(1, B, Cin, H, W)
    w = rnd(2, Cout, Cin, k, k) * (1.0 / np.sqrt(Cin * k * k))
    b = rnd(3, Cout, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, k // 2)
    x0, x1 = to_map(x[:, :C0], dev), (to_map(x[:, C0:], dev) if C1 else None)
    wd, bd = cl_weight(w).to(dev), b.to(dev)
    e_split = relerr(from_map(ops.conv_forward_split(x0, x1, wd, bd)), ref)
    e_fp32 = relerr(from_map(ops.conv_forward(x0, x1, wd, bd, allow_split=False)), ref)
    assert e_split < 2e-6 and e_split < 4 * e_fp32 + 2e-7, (e_split, e_fp32)
    gy = rnd(4, B, Cout, H, W)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, 1, k // 2).backward(gy.double())
    d0, d1 = ops.conv_dgrad_split(to_map(gy, dev), wd, C0, C1)
    got = torch.cat([from_map(d0)] + ([from_map(d1)] if C1 else []), 1)
    assert relerr(got, xr.grad) < 2e-6


@pytest.mark.parametrize("B,H,W,C0,C1,Cout,up", [(3, 64, 64, 64, 0, 64, False), (2, 64, 64, 64, 64, 64, True),
                                                  (1, 64, 128, 32, 32, 128, False), (600, 64, 64, 64, 0, 64, False),
                                                  (2, 64, 64, 128, 0, 128, False)])
def test_rows_kernel_2d_tiles_are_the_same_bits(dev, B, H, W, C0, C1, Cout, up, monkeypatch):
    """The unrolled 3x3 rows kernels with 2-D tiles (8 image rows x 16 pixels + a one-pixel halo, maps >= 64 wide) against
    the same kernels with whole-row tiles (RAC_ROWS_TILE2D=0): the arithmetic of an output pixel is untouched -- same K
    order, same operands -- so the results are bit-equal; forward with a fused epilogue and per-image scales, a second
    source, the decoder's half-resolution first source (a0_up), the persistent form (600 images), and the data gradient."""
    from robot_aware_control_amd import ops
    h, w_ = (H // 2, W // 2) if up else (H, W)
    x0 = to_map(rnd(51, B, C0, h, w_), dev)
    x1 = to_map(rnd(52, B, C1, H, W), dev) if C1 else None
    wt = cl_weight(rnd(53, Cout, C0 + C1, 3, 3) * 0.05).to(dev)
    scale, shift = (rnd(54, Cout).abs() + 0.5).to(dev), rnd(55, Cout, scale=0.2).to(dev)
    gy = to_map(rnd(56, B, Cout, H, W), dev)

    def run():
        y = ops.conv_forward_split(x0, x1, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, x0_up=up, per_image=True)
        raw = None if up else ops.conv_forward_split(x0, x1, wt, None)  # tensor-wide scale, no epilogue
        d0, d1 = (None, None) if up else ops.conv_dgrad_split(gy, wt, C0, C1)
        return y, ops.amax_tag(y).clone(), raw, d0, d1
    monkeypatch.setenv("RAC_ROWS_TILE2D", "1")
    got = run()
    monkeypatch.setenv("RAC_ROWS_TILE2D", "0")
    want = run()
    for a, b in zip(got, want):
        assert (a is None and b is None) or torch.equal(a, b)
    src = F.interpolate(from_map(x0).double(), scale_factor=2, mode="nearest") if up else from_map(x0).double()
    xin = torch.cat([src] + ([from_map(x1).double()] if C1 else []), 1)
    ref = F.leaky_relu(F.conv2d(xin, wt.cpu().double(), None, 1, 1) * scale.cpu().double().view(1, -1, 1, 1)
                       + shift.cpu().double().view(1, -1, 1, 1), 0.2)
    assert relerr(from_map(got[0]), ref) < 2e-6


@pytest.mark.parametrize("B,C0,C1,Cout", [(3, 64, 64, 256), (8, 128, 128, 512), (2, 96, 0, 128)])
def test_rows_kernel_5x5_padded_rows_are_the_same_bits(dev, B, C0, C1, Cout, monkeypatch):
    """5x5 convs on 16x16 maps (the ConvLSTM gate convs of a 128x128 model, BASELINE configs[4]): the padded-row form of
    the rows kernel (a tap that leaves the image row reads zeros by address; RAC_ROWS_FAST5, default) against the generic
    loop with its per-lane tap masks (RAC_ROWS_FAST5=0): same operands, same order of sums -- bit-equal; forward (two
    sources, K split and unsplit with per-image scales), the data gradient, and against fp64."""
    from robot_aware_control_amd import ops
    H = W = 16
    x0 = to_map(rnd(71, B, C0, H, W), dev)
    x1 = to_map(rnd(72, B, C1, H, W), dev) if C1 else None
    wt = cl_weight(rnd(73, Cout, C0 + C1, 5, 5) * 0.03).to(dev)
    gy = to_map(rnd(74, B, Cout, H, W), dev)

    def run():
        y = ops.conv_forward_split(x0, x1, wt, None, per_image=True)
        slabs, n, stride = ops.conv_forward_split(x0, x1, wt, want_slabs=True)
        d0, d1 = ops.conv_dgrad_split(gy, wt, C0, C1)
        return y, slabs.clone(), d0, d1
    monkeypatch.setenv("RAC_ROWS_FAST5", "1")
    got = run()
    monkeypatch.setenv("RAC_ROWS_FAST5", "0")
    want = run()
    for a, b in zip(got, want):
        assert (a is None and b is None) or torch.equal(a, b)
    xin = torch.cat([from_map(x0).double()] + ([from_map(x1).double()] if C1 else []), 1)
    assert relerr(from_map(got[0]), F.conv2d(xin, wt.cpu().double(), None, 1, 2)) < 2e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout,fused", [
    (3, 64, 64, 64, 64, True),      # 2-D tiles, 64 columns
    (600, 64, 64, 64, 64, True),    # the persistent form
    (2, 64, 64, 128, 128, True),    # 2-D tiles, 128 columns
    (4, 32, 32, 128, 128, True),    # whole rows, two 16-pixel blocks per image row
    (4, 32, 32, 64, 32, False),     # 32-column kernel: two row blocks per wave cannot hold both rows of a pair
    (8, 16, 16, 256, 256, True),    # whole rows, one block per image row, two column tiles
    (8, 16, 16, 128, 96, False),    # 96 columns do not fill the column blocks: separate pass
    (3, 8, 8, 256, 512, False),     # a map that fits a tile: separate pass
])
def test_conv_pools_in_its_epilogue_with_the_same_bits(dev, B, H, W, Cin, Cout, fused, monkeypatch):
    """conv_forward_split(pool=True) = (y, MaxPool2d(2)(y)) of a frozen vgg layer (vgg_64.py:8-18, 104-129): where the conv
    kernel can pool in its epilogue (rac_conv2d_fwd_split_pool_ok) the pooled map comes out of the same launch, elsewhere
    from rac_maxpool2_fwd -- the same bits either way, y untouched, and the pooled map is tagged with y's per-image maxima."""
    from robot_aware_control_amd import _lib, ops
    x = to_map(rnd(71, B, Cin, H, W), dev)
    wt = cl_weight(rnd(72, Cout, Cin, 3, 3) * 0.05).to(dev)
    scale, shift = (rnd(73, Cout).abs() + 0.5).to(dev), rnd(74, Cout, scale=0.2).to(dev)
    launches = []
    real = _lib.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (launches.append(name), real(name, *a))[1])
    ops._POOL_OK.clear()
    y, yp = ops.conv_forward_split(x, None, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True, pool=True)
    assert ("rac_maxpool2_fwd" not in launches) == fused, launches
    y0 = ops.conv_forward_split(x, None, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True)
    want = ops.MaxPool2.apply(y0)
    assert torch.equal(y, y0) and torch.equal(yp, want)
    assert torch.equal(ops.amax_tag(yp), ops.amax_tag(y)) and torch.equal(ops.amax_tag(y), ops.amax_tag(y0))
    ref = F.max_pool2d(from_map(y0), 2)
    assert torch.equal(from_map(yp), ref)
    monkeypatch.setattr(ops, "POOL_FUSED", False)  # the switch: always the separate pass
    ops._POOL_OK.clear()
    y2, yp2 = ops.conv_forward_split(x, None, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True, pool=True)
    assert torch.equal(y2, y0) and torch.equal(yp2, want)
    ops._POOL_OK.clear()
    if fused and W == 64:
        # a cached "yes" that the library no longer honours (whole-row tiles forced: a 64-wide map's two rows per tile sit in
        # different waves): the launch is refused, the separate pass takes over, same bits
        monkeypatch.setattr(ops, "POOL_FUSED", True)
        ops.conv_forward_split(x, None, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True, pool=True)
        assert all(ops._POOL_OK.values())
        monkeypatch.setenv("RAC_ROWS_TILE2D", "0")
        y3, yp3 = ops.conv_forward_split(x, None, wt, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True,
                                         pool=True)
        assert torch.equal(y3, y0) and torch.equal(yp3, want) and not any(ops._POOL_OK.values())
        ops._POOL_OK.clear()


@pytest.mark.parametrize("B,Cin,Cout", [(1, 64, 64), (2, 128, 64)])
def test_rows_kernel_takes_128_wide_maps(dev, B, Cin, Cout):
    """128x128 maps (the 64-channel layers of a 128x128 model, BASELINE configs[4]): no whole-row tile exists (a row's halo
    alone is 256 pixels), the 2-D tiles take them -- forward, data gradient and weight gradient on the split pipe, against
    fp64."""
    from robot_aware_control_amd import ops
    H = W = 128
    assert ops.split_supported(H, W, 3, Cin, Cout, 0)
    x, w = rnd(61, B, Cin, H, W), rnd(62, Cout, Cin, 3, 3) * (1.0 / np.sqrt(Cin * 9))
    xd, wd = to_map(x, dev), cl_weight(w).to(dev)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert relerr(from_map(ops.conv_forward_split(xd, None, wd)), ref) < 2e-6
    gy = rnd(63, B, Cout, H, W)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xr, wr, None, 1, 1).backward(gy.double())
    d0, _ = ops.conv_dgrad_split(to_map(gy, dev), wd, Cin, 0)
    assert relerr(from_map(d0), xr.grad) < 2e-6
    wd.grad = torch.zeros_like(wd)
    ops.conv_wgrad_split_acc(to_map(gy, dev), xd, None, wd)
    assert relerr(wd.grad.cpu(), wr.grad) < 2e-6


@pytest.mark.parametrize("case", [(2, 8, 8, 128, 128, 256, 5), (3, 8, 8, 128, 0, 160, 3), (1, 16, 16, 64, 0, 96, 3),
                                  (5, 4, 8, 32, 0, 64, 3), (1, 16, 16, 64, 0, 128, 3), (4, 8, 8, 128, 128, 384, 3),
                                  (2, 32, 32, 64, 64, 64, 3), (3, 16, 16, 128, 0, 64, 5), (2, 8, 8, 64, 0, 48, 3),
                                  (2, 64, 64, 64, 0, 64, 3), (1, 32, 32, 128, 0, 128, 3), (3, 32, 32, 96, 0, 96, 3),
                                  (1, 64, 64, 64, 64, 64, 3),
                                  # 6- / 12-row maps (48x64 frames): a 32-row group is not whole images, a lane's image row
                                  # changes from group to group; 7 images of 6 rows = 42 rows: the second group is ragged
                                  (7, 6, 8, 64, 64, 96, 5), (3, 12, 16, 64, 0, 64, 3), (11, 6, 8, 128, 128, 128, 3)])
@pytest.mark.parametrize("presplit", [False, True])
def test_wgrad_split_precision(dev, case, presplit, monkeypatch):
    """Weight gradient on the split-precision pipe against fp64, next to the exact-fp32 MFMA kernel; accumulation into
    .grad; the deferred (time-batched) form.  `presplit`: the operands are split into their fp16 parts once by
    rac_split_steps (what the ConvLSTM gate weights get) instead of inside the kernel.  The 3x3 cases with Cout <= 128 on
    32x32 / 64x64 maps take the all-taps kernel (wgrad16_allky_kernel) unless presplit."""
    from robot_aware_control_amd import ops
    monkeypatch.setattr(ops, "WGRAD_PRESPLIT_MIN_READERS", 1 if presplit else 10 ** 9)
    B, H, W, C0, C1, Cout, k = case
    Cin = C0 + C1
    x = rnd(1, B, Cin, H, W)
    w = (rnd(2, Cout, Cin, k, k) * 0.02).requires_grad_(True)
    gy = rnd(4, B, Cout, H, W) * 1e-4
    F.conv2d(x.double(), w.double(), None, 1, k // 2).backward(gy.double())  # fp64 reference... via float weight
    ref = w.grad.double()
    x0, x1 = to_map(x[:, :C0], dev), (to_map(x[:, C0:], dev) if C1 else None)
    wd = cl_weight(w.detach()).to(dev).requires_grad_(True)
    ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wd)
    e_split = relerr(wd.grad.cpu(), ref)
    wd2 = cl_weight(w.detach()).to(dev).requires_grad_(True)
    ops.conv_wgrad_acc(to_map(gy, dev), x0, x1, wd2)
    e_fp32 = relerr(wd2.grad.cpu(), ref)
    assert e_split < 3e-6 and e_split < 4 * e_fp32 + 3e-7, (e_split, e_fp32)
    ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wd)  # accumulates
    assert relerr(wd.grad.cpu(), 2 * ref) < 3e-6
    if k == 3 and Cout <= 128 and H % 32 == 0 and not presplit:
        # the all-taps form against the kernel-row form: same sums, another association
        monkeypatch.setattr(ops, "WGRAD_ALLKY", False)
        wrow = cl_weight(w.detach()).to(dev).requires_grad_(True)
        ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wrow)
        monkeypatch.setattr(ops, "WGRAD_ALLKY", True)
        wall = cl_weight(w.detach()).to(dev).requires_grad_(True)
        ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wall)
        assert relerr(wall.grad.cpu(), ref) < 3e-6 and relerr(wrow.grad.cpu(), ref) < 3e-6
        assert relerr(wall.grad.cpu(), wrow.grad.cpu().double()) < 2e-6
    # no atomics: the same operands give the same bits, K split (slabs + fixed-order accumulate) included
    for forced in (None, "3"):
        reps = []
        for _ in range(2):
            wr = cl_weight(w.detach()).to(dev).requires_grad_(True)
            if forced:
                monkeypatch.setenv("RAC_WGRAD_SPLITK", forced)
            ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wr)
            monkeypatch.delenv("RAC_WGRAD_SPLITK", raising=False)
            reps.append(wr.grad.clone())
        assert torch.equal(reps[0], reps[1]) and relerr(reps[0].cpu(), ref) < 3e-6
    # deferred: the time steps' operands in ONE launch when the context exits
    wd3 = cl_weight(w.detach()).to(dev).requires_grad_(True)
    with ops.deferred_wgrad():
        ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wd3, defer=True)
        ops.conv_wgrad_split_acc(to_map(2 * gy, dev), x0, x1, wd3, defer=True)
        ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wd3, defer=True)
        assert wd3.grad is None or float(wd3.grad.abs().max()) == 0.0
    assert relerr(wd3.grad.cpu(), 4 * ref) < 3e-6
    if C1:
        # a step whose second source is flagged all-zero (a ConvLSTM's first step) is skipped in the x1 half of dw:
        # same result as computing with the zeros, whatever position the step has in the batch
        z1 = torch.zeros_like(x1)
        z1._rac_zero = True
        xz = x.clone()
        xz[:, C0:] = 0
        wz = w.detach().clone().requires_grad_(True)
        F.conv2d(xz.double(), wz.double(), None, 1, k // 2).backward(gy.double())
        ref_z = ref + 2 * wz.grad.double()
        wd4 = cl_weight(w.detach()).to(dev).requires_grad_(True)
        with ops.deferred_wgrad():
            ops.conv_wgrad_split_acc(to_map(gy, dev), x0, z1, wd4, defer=True)
            ops.conv_wgrad_split_acc(to_map(gy, dev), x0, x1, wd4, defer=True)
            ops.conv_wgrad_split_acc(to_map(gy, dev), x0, z1, wd4, defer=True)
        assert relerr(wd4.grad.cpu(), ref_z) < 3e-6
        wd5 = cl_weight(w.detach()).to(dev).requires_grad_(True)
        ops.conv_wgrad_split_acc(to_map(gy, dev), x0, z1, wd5)  # every step zero: the x1 half stays untouched
        assert relerr(wd5.grad.cpu(), wz.grad.double()) < 3e-6 and float(wd5.grad[:, C0:].abs().max()) == 0.0


def test_compiled_cpp_consumer_calls_the_library(dev):
    """tests/abi_consumer.cpp: a C++ host compiled against include/rac_hip.h and linked with -lrac_hip (no Python, no torch)
    makes one rac_conv2d call on its own hipMalloc'ed buffers and stream, checks it against a host loop, and gets
    RAC_EINVAL + a message for a bad argument."""
    import subprocess
    import __graft_entry__ as entry
    exe = entry.build_abi_consumer()
    res = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    kv = dict(line.split() for line in res.stdout.splitlines())
    assert float(kv["conv_max_rel_err"]) < 1e-5 and int(kv["bad_ksize_rc"]) == -1


def test_amax_tag_does_not_survive_gradient_accumulation(dev):
    """A conv output with TWO consumers: autograd sums their gradients, possibly in place into the first one's buffer,
    whose max-|x| tag (left by the data-gradient kernel that produced it) then under-estimates the sum.  The tag carries
    the tensor version it was measured at, so the accumulated gradient is measured again: dx and dW stay at fp32 level
    even when the second gradient is 300 times the first."""
    from robot_aware_control_amd import ops
    B, H, W, C = 4, 8, 8, 128
    x = to_map(rnd(71, B, C, H, W), dev).requires_grad_(True)
    ws = [cl_weight(rnd(72 + i, C, C, 3, 3) * 0.05).to(dev).requires_grad_(True) for i in range(3)]
    bs = [rnd(75 + i, C, scale=0.1).to(dev).requires_grad_(True) for i in range(3)]
    y = ops.ConvBias.apply(x, None, ws[0], bs[0], ops.ACT_NONE)
    a = ops.ConvBias.apply(y, None, ws[1], bs[1], ops.ACT_NONE)
    b = ops.ConvBias.apply(y, None, ws[2], bs[2], ops.ACT_NONE)
    ga, gb = to_map(rnd(78, B, C, H, W), dev), to_map(rnd(79, B, C, H, W) * 300.0, dev)
    torch.autograd.backward([a, b], [ga, gb])
    xr = from_map(x.detach()).double().requires_grad_(True)
    wr = [w.detach().cpu().double().requires_grad_(True) for w in ws]
    yr = F.conv2d(xr, wr[0], bs[0].detach().cpu().double(), 1, 1)
    ar = F.conv2d(yr, wr[1], bs[1].detach().cpu().double(), 1, 1)
    br = F.conv2d(yr, wr[2], bs[2].detach().cpu().double(), 1, 1)
    torch.autograd.backward([ar, br], [from_map(ga).double(), from_map(gb).double()])
    assert relerr(from_map(x.grad), xr.grad) < 2e-6
    assert relerr(ws[0].grad.cpu(), wr[0].grad) < 2e-6
    # the mechanism itself: an in-place change invalidates the tag
    t = ops.tag_amax(x.detach().clone(), ops.amax_slot(dev))
    assert ops.amax_tag(t) is not None
    t.add_(1.0)
    assert ops.amax_tag(t) is None and int(ops.amax_for(t).item()) == int(t.abs().max().view(torch.int32).item())


def test_fused_adam_step_writes_the_next_parts(dev):
    """rac_adam_frag_multi + rac_adam_ranges against rac_adam_step + the separate refresh: the same p / m / v bits, the
    fragment parts of the UPDATED weights under the bound's scale (bit-equal to rac_weight_frag_split run on them with
    that slot), the exact new maxima kept for the next bound; taken only when every registered weight is current."""
    from robot_aware_control_amd import _lib, ops
    shapes = [(64, 96, 3), (128, 64, 5), (32, 32, 3)]
    sizes = [co * ci * k * k for co, ci, k in shapes]
    gaps = [8, 20, 12, 36]  # "biases" between and around the conv weights (multiples of 4 floats)
    total = sum(sizes) + sum(gaps)
    g0 = torch.Generator().manual_seed(5)
    flat = (torch.randn(total, generator=g0) * 0.05).to(dev)
    grad = (torch.randn(total, generator=g0) * 0.01).to(dev)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    ws, off = [], gaps[0]
    for (co, ci, k), n, gap in zip(shapes, sizes, gaps[1:]):
        ws.append(torch.as_strided(flat, (co, ci, k, k), (k * k * ci, 1, k * ci, ci), off))
        off += n + gap
    assert not ops.fused_adam_step(flat, grad, m, v, 1e-3, 0.9, 0.999, 1e-8, 1)  # nothing registered yet
    for w, both in zip(ws, (True, True, False)):
        ops.weight_parts(w)
        if both:
            ops.weight_parts(w, transposed=True)
    ref = [t.clone() for t in (flat, m, v)]
    for step in (1, 2, 3):
        lr = 1e-3
        _lib.call("rac_adam_step", ref[0].data_ptr(), grad.data_ptr(), ref[1].data_ptr(), ref[2].data_ptr(), total, lr, 0.9,
                  0.999, 1e-8, step, _lib.stream_ptr())
        assert ops.fused_adam_step(flat, grad, m, v, lr, 0.9, 0.999, 1e-8, step)
        assert torch.equal(flat, ref[0]) and torch.equal(m, ref[1]) and torch.equal(v, ref[2])
        st = ops._wp_state(dev)
        for w, both in zip(ws, (True, True, False)):
            ent = ops._WP_ENTRIES[id(w)]
            assert ent.tag == ops._wp_tag(w)  # no lazy refresh will follow
            exact = st["exact"][ent.idx:ent.idx + 1]
            assert int(exact.item()) == int(w.abs().max().view(torch.int32).item())
            bound = ent.slot.view(torch.float32)
            assert float(w.abs().max()) <= float(bound) <= float(w.abs().max()) + 10 * lr
            co, ci, k, _ = w.shape
            for transposed in ((False, True) if both else (False,)):
                want = torch.empty_like(ent.parts[transposed])
                _lib.call("rac_weight_frag_split", w.data_ptr(), ent.slot.data_ptr(), want.data_ptr(), co, ci, k,
                          1 if transposed else 0, w.numel(), _lib.stream_ptr())
                assert torch.equal(ent.parts[transposed], want), (tuple(w.shape), transposed)
            # and the conv that consumes them is right
            x = to_map(rnd(91, 2, ci, 8, 8), dev)
            got = ops.conv_forward_split(x, None, w, None)
            refc = F.conv2d(from_map(x).double(), w.detach().cpu().double(), None, 1, k // 2)
            assert relerr(from_map(got), refc) < 2e-6
        grad.mul_(1.7)
    # `late`: the weights behind an offset (here the second and third) are updated on a side stream; after the returned
    # event the buffers and the parts are the bits of the one-stream pass
    keep = [t.clone() for t in (flat, m, v)]
    assert ops.fused_adam_step(flat, grad, m, v, 1e-3, 0.9, 0.999, 1e-8, 4) is True
    want = [t.clone() for t in (flat, m, v)] + [q.clone() for w in ws for q in ops._WP_ENTRIES[id(w)].parts.values()]
    want_slots = [ops._WP_ENTRIES[id(w)].slot.clone() for w in ws]
    for t, k0 in zip((flat, m, v), keep):
        t.copy_(k0)
    for w in ws:  # (the restored weights' parts: the lazy path, as after load_state_dict)
        ops.weight_parts(w)
    off1 = (ws[1].data_ptr() - flat.data_ptr()) // 4
    # (two late groups, in the order their consumers run: each its own launch and event on the side stream)
    res = ops.fused_adam_step(flat, grad, m, v, 1e-3, 0.9, 0.999, 1e-8, 4,
                              late=(off1, [frozenset([ws[2].data_ptr()]), frozenset([ws[1].data_ptr()])]))
    span = lambda w: ((w.data_ptr() - flat.data_ptr()) // 4, w.numel())
    assert res is not True and [sp for _, sp in res] == [[span(ws[2])], [span(ws[1])]]
    torch.cuda.current_stream().wait_event(res[-1][0])  # (one in-order stream: the last event covers every group)
    got = [flat, m, v] + [q for w in ws for q in ops._WP_ENTRIES[id(w)].parts.values()]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    assert all(torch.equal(ops._WP_ENTRIES[id(w)].slot, sl) for w, sl in zip(ws, want_slots))
    # a weight changed behind the parts' back (load_state_dict ...): the fused step declines, the lazy refresh repairs
    ws[0].mul_(2.0)
    assert not ops.fused_adam_step(flat, grad, m, v, 1e-3, 0.9, 0.999, 1e-8, 5)
    ops.weight_parts(ws[0])
    assert ops.fused_adam_step(flat, grad, m, v, 1e-3, 0.9, 0.999, 1e-8, 5)


def test_bound_scaled_weight_parts_cost_at_most_one_bit(dev):
    """After a fused optimiser step the weight parts are scaled by the exponent of an UPPER BOUND of max |w| (old maximum
    + the largest step Adam can take, ops.adam_step_bound), not of the exact maximum: when the bound crosses a power of two
    that the maximum does not, the operands keep 21 of their 22 significant bits.  Pinned here on the realistic worst case
    -- small weights (max |w| = 0.0156 = just under 2^-6, a gate conv with large fan-in) and lr 1e-3, whose margin 7.3e-3
    lifts the bound over 2^-6: the conv with bound-scaled parts stays within 2e-6 of fp64 (the bar of every split kernel)
    and within 2.5x the error of the exact-scaled parts; at most ONE exponent step separates the two scales."""
    from robot_aware_control_amd import _lib, ops
    co, ci, k = 128, 256, 3
    w = (rnd(33, co, ci, k, k) * 0.004).clamp_(-0.0156, 0.0156)
    w[3, 5, 1, 1] = 0.0156
    w = cl_weight(w).to(dev)
    assert w.stride() == (k * k * ci, 1, k * ci, ci)
    x = to_map(rnd(34, 4, ci, 8, 8), dev)
    refc = F.conv2d(from_map(x).double(), w.detach().cpu().double(), None, 1, 1)
    exact_bits = w.abs().max().view(torch.int32).reshape(1).clone()
    bound = (w.abs().max() + 1e-3 * ops.adam_step_bound(0.9, 0.999) * 1.01).reshape(1)
    assert float(bound) > 2.0 ** -6 > float(w.abs().max())          # the bound crosses the power of two
    errs = {}
    for name, slot in (("exact", exact_bits), ("bound", bound.view(torch.int32).clone())):
        parts = torch.empty((2, w.numel()), device=dev, dtype=torch.float16)
        _lib.call("rac_weight_frag_split", w.data_ptr(), slot.data_ptr(), parts.data_ptr(), co, ci, k, 0, w.numel(),
                  _lib.stream_ptr())
        out = torch.empty((4, 8, 8, co), device=dev, dtype=torch.float32)
        ops._split_launch(x, None, ops.amax_of(x), None, parts, slot, out, B=4, H=8, W=8, k=k, Cin=ci, Cout=co, C0=ci)
        errs[name] = relerr(from_map(out), refc)
    assert errs["bound"] < 2e-6 and errs["bound"] <= 2.5 * errs["exact"] + 1e-7, errs
    import math
    assert math.frexp(float(bound))[1] - math.frexp(float(w.abs().max()))[1] == 1


@pytest.mark.parametrize("B,H,W,g,k", [(5, 8, 8, 64, 5), (3, 6, 8, 128, 3), (2, 8, 8, 512, 5)])
def test_frozen_cell_in_the_gate_conv_epilogue(dev, B, H, W, g, k, monkeypatch):
    """rac_convlstm_cell_fwd_split (the frozen model's ConvLSTM cell in one launch, gate-interleaved weight rows) against
    the two-launch form (gate conv slabs + rac_lstm_cell_fwd): same pre-activations to the bit, transcendentals to ~1e-7
    absolute; against fp64; zero initial state (hidden half of K skipped); batch-invariant to the bit."""
    from robot_aware_control_amd import ops
    x = to_map(rnd(81, B, g, H, W), dev)
    hp = ops.tag_amax(to_map(torch.tanh(rnd(82, B, g, H, W)), dev), ops.amax_one(dev))
    cp = to_map(rnd(83, B, g, H, W), dev)
    wt = cl_weight(rnd(84, 4 * g, 2 * g, k, k) * (1.0 / np.sqrt(2 * g * k * k))).to(dev)
    bias = rnd(85, 4 * g, scale=0.1).to(dev)
    assert ops.fused_cell_ok(x, wt)
    with torch.no_grad():
        h1, c1 = ops.LstmCell.apply(x, hp, cp, wt, bias, False)
        monkeypatch.setattr(ops, "FUSED_CELL", False)
        h0, c0 = ops.LstmCell.apply(x, hp, cp, wt, bias, False)
        monkeypatch.setattr(ops, "FUSED_CELL", True)
    assert float((h1 - h0).abs().max()) < 6e-7 and float((c1 - c0).abs().max()) < 6e-7 * max(1.0, float(c0.abs().max()))
    gates = F.conv2d(torch.cat([from_map(x), from_map(hp)], 1).double(), wt.cpu().double(), bias.cpu().double(), 1, k // 2)
    i, f, o, gg = gates.chunk(4, 1)
    c_ref = torch.sigmoid(f) * from_map(cp).double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    assert relerr(from_map(c1), c_ref) < 6e-6 and relerr(from_map(h1), h_ref) < 6e-6  # (fp32 sums over K up to 25 600)
    with torch.no_grad():
        sub = ops.LstmCell.apply(x[B - 1:].contiguous(), ops.tag_amax(hp[B - 1:].contiguous(), ops.amax_one(dev)),
                                 cp[B - 1:].contiguous(), wt, bias, False)
        assert torch.equal(sub[0], h1[B - 1:]) and torch.equal(sub[1], c1[B - 1:])
        z0 = torch.zeros_like(x)
        z0._rac_zero = True
        ops.tag_amax(z0, ops.amax_one(dev))
        hz, cz = ops.LstmCell.apply(x, z0, z0, wt, bias, False)
    gz = F.conv2d(torch.cat([from_map(x), torch.zeros_like(from_map(x))], 1).double(), wt.cpu().double(), bias.cpu().double(), 1, k // 2)
    i, f, o, gg = gz.chunk(4, 1)
    cz_ref = torch.sigmoid(i) * torch.tanh(gg)
    assert relerr(from_map(cz), cz_ref) < 6e-6 and relerr(from_map(hz), torch.sigmoid(o) * torch.tanh(cz_ref)) < 6e-6


@pytest.mark.parametrize("M,Cc,G", [(1024, 512, 1), (5 * 1024, 512, 5), (2 * 4096, 64, 2), (3 * 768, 256, 3)])
def test_bn_apply_act_is_finalize_plus_affine(dev, monkeypatch, M, Cc, G):
    """rac_bn_apply_act (BatchNorm finalize + affine + LeakyReLU in one launch) against rac_bn_finalize + rac_affine_act:
    the same bits in the activated map, in scale / shift / mean / invstd, in the running statistics (two momentum updates
    per group, groups in order) and in the map's maximum."""
    from robot_aware_control_amd import ops
    raw = (rnd(3, M, Cc) * 1.7 + 0.3).to(dev).view(G, M // G, 1, Cc)
    gamma, beta = (1 + rnd(4, Cc, scale=0.1)).to(dev), rnd(5, Cc, scale=0.1).to(dev)
    stats = torch.zeros(G, 2, Cc, device=dev, dtype=torch.float64)
    ops.call("rac_col_stats", ops.ptr(raw), ops.ptr(stats), M, Cc, G, ops.stream_ptr())
    out = {}
    for fused in (False, True):
        monkeypatch.setattr(ops, "BN_FUSED_APPLY", fused)
        rm, rv = rnd(6, Cc, scale=0.1).to(dev), (1 + rnd(7, Cc, scale=0.1).abs()).to(dev)
        y, aff = ops.bn_apply_act(raw, stats, gamma, beta, rm, rv, 2, M, Cc, G)
        torch.cuda.synchronize()
        out[fused] = (y.clone(), aff.clone(), rm.clone(), rv.clone(), ops.amax_tag(y).clone())
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
    y, aff = out[True][0], out[True][1]
    x = raw.double().view(G, M // G, Cc)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    ref = (x - mean[:, None]) / torch.sqrt(var[:, None] + 1e-5) * gamma.double() + beta.double()
    ref = torch.where(ref > 0, ref, 0.2 * ref)
    assert float((y.double().view(G, M // G, Cc) - ref).abs().max()) < 1e-4
    assert int(out[True][4].item()) == int(y.abs().max().view(torch.int32).item())


@pytest.mark.parametrize("B,H,W,g", [(5, 6, 8, 256), (3, 8, 8, 64), (2, 8, 8, 512)])
def test_norm_lstm_cell_fused_forward(dev, monkeypatch, B, H, W, g):
    """rac_norm_lstm_cell_fwd (the frozen NormConvLSTMCell behind its gate convs in one launch) against torch's GroupNorm /
    sigmoid / tanh in fp64 (lstm.py:174-198) and against this package's unfused kernels; an image's result is the same bits
    alone and in a batch."""
    from robot_aware_control_amd import ops
    g_ih = (rnd(1, B, H, W, 4 * g) * 1.3 + 0.2).to(dev)
    g_hh = (rnd(2, B, H, W, 4 * g) * 0.7 - 0.1).to(dev)
    c_prev = rnd(3, B, H, W, g).to(dev)
    gn = [((1 + rnd(10 + i, n, scale=0.1)).to(dev), rnd(20 + i, n, scale=0.1).to(dev)) for i, n in enumerate((4 * g, 4 * g, g))]
    h, c = ops.norm_cell_frozen(g_ih, g_hh, c_prev, *gn)
    planes = lambda t: t.permute(0, 3, 1, 2).double().cpu()
    gates = (F.group_norm(planes(g_ih), 16, gn[0][0].double().cpu(), gn[0][1].double().cpu(), 1e-5)
             + F.group_norm(planes(g_hh), 16, gn[1][0].double().cpu(), gn[1][1].double().cpu(), 1e-5))
    i_, f_, o_, gg = gates.chunk(4, 1)
    c_ref = F.group_norm(torch.sigmoid(f_) * planes(c_prev) + torch.sigmoid(i_) * torch.tanh(gg), 16,
                         gn[2][0].double().cpu(), gn[2][1].double().cpu(), 1e-5)
    h_ref = torch.sigmoid(o_) * torch.tanh(c_ref)
    assert relerr(planes(c), c_ref) < 2e-6 and relerr(planes(h), h_ref) < 2e-6
    # the unfused kernels (GroupNorm x 3, cell core, output)
    with torch.no_grad():
        n_ih = ops.GroupNorm.apply(g_ih, gn[0][0], gn[0][1], 16)
        n_hh = ops.GroupNorm.apply(g_hh, gn[1][0], gn[1][1], 16)
        c_raw, act = ops.NormCellCore.apply(n_ih, n_hh, c_prev)
        c2 = ops.GroupNorm.apply(c_raw, gn[2][0], gn[2][1], 16)
        h2 = ops.LstmOut.apply(act, c2)
    assert relerr(c, c2) < 2e-6 and relerr(h, h2) < 2e-6
    h1, c1 = ops.norm_cell_frozen(g_ih[1:2].contiguous(), g_hh[1:2].contiguous(), c_prev[1:2].contiguous(), *gn)
    assert torch.equal(h1[0], h[1]) and torch.equal(c1[0], c[1])


@pytest.mark.parametrize("B,H,W,g", [(4, 6, 8, 256), (2, 8, 8, 64)])
def test_norm_lstm_cell_one_node_matches_seven(dev, monkeypatch, B, H, W, g):
    """ops.NormLstmCell (the training NormConvLSTMCell as one autograd node around rac_norm_lstm_cell_fwd) against the
    seven-node form (ConvBias x 2, GroupNorm x 3, NormCellCore, LstmOut): h, c and every gradient -- inputs, both conv
    weights and biases, the three norms' affines -- over two chained steps (the second feeds on the first's state)."""
    import argparse
    from robot_aware_control_amd import model as mdl, ops
    out = {}
    for node, fused_bwd in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(ops, "NORM_CELL_NODE", node)
        monkeypatch.setattr(ops, "NORM_CELL_BWD_FUSED", fused_bwd)
        torch.manual_seed(0)
        cell = mdl._NormLstmCell(g, 5).to(dev)
        with torch.no_grad():
            for i, p in enumerate(cell.parameters()):
                src = rnd(40 + i, *p.shape, scale=(1.0 / np.sqrt(g * 25)) if p.dim() == 4 else 0.1)
                p.copy_((src if p.dim() != 1 or "norm" not in "" else src).to(dev))
            for gn in (cell.ih_gates[1], cell.hh_gates[1], cell.c_norm):
                gn.weight.add_(1.0)
        x1 = rnd(1, B, H, W, g).to(dev).requires_grad_(True)
        x2 = rnd(2, B, H, W, g).to(dev).requires_grad_(True)
        h0 = (rnd(3, B, H, W, g) * 0.5).to(dev).requires_grad_(True)
        c0 = rnd(4, B, H, W, g).to(dev).requires_grad_(True)
        with ops.deferred_wgrad():
            h1, c1 = cell(x1, (ops.tag_amax(h0, ops.amax_of(h0)), c0))
            h2, c2 = cell(x2, (h1, c1))
            loss_h, loss_c = rnd(5, B, H, W, g).to(dev), rnd(6, B, H, W, g).to(dev)
            torch.autograd.backward([h2, c2, h1], [loss_h, loss_c, 0.3 * loss_h])
        torch.cuda.synchronize()
        out[(node, fused_bwd)] = ([h1, c1, h2, c2, x1.grad, x2.grad, h0.grad, c0.grad]
                                   + [p.grad for p in cell.parameters()])
    for key in ((True, True), (True, False)):
        for i, (a, b) in enumerate(zip(out[key], out[(False, False)])):
            assert relerr(a, b) < 5e-6, (key, i)


@pytest.mark.parametrize("shape", [(16, 8, 8, 256, 0, 512), (16, 16, 16, 128, 128, 256), (16, 6, 8, 256, 0, 512), (4, 16, 16, 64, 0, 128)])
def test_small_bn_layer_one_launch_each_way(dev, monkeypatch, shape):
    """rac_bn_small_fwd / rac_bn_small_bwd (one time step's small train-mode vgg layer: split-K combine + statistics + apply,
    and reduce + apply, in ONE launch each -- a workgroup owns a slice of channels and all rows) against the multi-launch
    forms: the activated map, the running statistics, every gradient to 2e-6 (the sums run in another order), and the
    single-launch forms twice in a row to the bit."""
    from robot_aware_control_amd import ops
    B, H, W, C0, C1, Cout = shape
    Cin = C0 + C1
    x = rnd(1, B, Cin, H, W)
    w = rnd(2, Cout, Cin, 3, 3) * (1.4 / np.sqrt(Cin * 9))
    gy = rnd(7, B, Cout, H, W)
    out = {}
    for key, small in (("small", True), ("again", True), ("multi", False)):
        monkeypatch.setattr(ops, "BN_SMALL", small)
        x0 = to_map(x[:, :C0], dev).requires_grad_(True)
        x1 = to_map(x[:, C0:], dev).requires_grad_(True) if C1 else None
        wd = cl_weight(w.clone()).to(dev).requires_grad_(True)
        gd, bd = (1 + rnd(3, Cout, scale=0.1)).to(dev).requires_grad_(True), rnd(4, Cout, scale=0.1).to(dev).requires_grad_(True)
        rm, rv = rnd(5, Cout, scale=0.1).to(dev), (1 + rnd(6, Cout, scale=0.1).abs()).to(dev)
        y = ops.VggLayer.apply(x0, x1, wd, gd, bd, rm, rv, True, 2, None)
        y.backward(to_map(gy, dev))
        torch.cuda.synchronize()
        out[key] = [y.detach(), rm, rv, x0.grad, wd.grad, gd.grad, bd.grad] + ([x1.grad] if C1 else []) + [ops.amax_tag(y)]
    assert bool(ops._lib.load().rac_bn_small_ok(B * H * W, Cout))
    for i, (a, b) in enumerate(zip(out["small"], out["again"])):
        assert torch.equal(a, b), i
    for i, (a, b) in enumerate(zip(out["small"][:-1], out["multi"][:-1])):
        assert relerr(a, b) < 2e-6, i
    assert int(out["small"][-1].item()) == int(out["small"][0].abs().max().view(torch.int32).item())

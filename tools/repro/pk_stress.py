"""Multi-process stress of the frozen model's first-layer kernel (rac_first_layer_fwd): the check that caught the
v_pk_fma_f32 miscompute of round 2 ("A correctness note"), kept as a tool.

    RAC_HIP_LIB=<lib> python tools/repro/pk_stress.py <processes> <launches>

Every process runs the kernel `launches` times on one fixed input and compares each output with its first one bit for bit
(same input, same kernel: any difference is a wrong result); differing elements are attributed to the lane of the wave
that computed them.  Run with the shipped library (scalar v_fma_f32) and with a build that keeps the compiler's packed
form (tools/build_variant.sh with RAC_PACKED=1)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, launches, q):
    from robot_aware_control_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, H, W, Cm = 256, 64, 64, 2
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(B, Cm, H, W, generator=g) > 0.7).float().to(dev)
    zm = (torch.rand(B, 1, H, W, generator=g) > 0.5).float().to(dev)
    w = (torch.randn(64, 3 + Cm, 3, 3, generator=g) * 0.2).contiguous(memory_format=torch.channels_last).to(dev)
    scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    first = ops.first_layer_frozen(img, zm, mask, w, scale, shift).clone()
    bad_launches, lanes = 0, torch.zeros(64, dtype=torch.long, device=dev)
    for _ in range(launches):
        out = ops.first_layer_frozen(img, zm, mask, w, scale, shift)
        if not torch.equal(out, first):
            bad_launches += 1
            b, y, x, co = torch.nonzero(out != first, as_tuple=True)
            # first_layer_kernel: 16x16 pixel tile per workgroup; thread = (pixel quad = (y % 16) * 4 + (x % 16) / 4) * 4
            # + co / 16; lane = thread & 63
            tid = (((y % 16) * 4 + (x % 16) // 4) * 4 + co // 16)
            lanes += torch.bincount(tid % 64, minlength=64)
    lq = lanes.view(4, 16).sum(1).tolist()
    q.put((rank, bad_launches, lq))


def main():
    procs, launches = int(sys.argv[1]), int(sys.argv[2])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, launches, q)) for r in range(procs)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=900) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    lib = os.environ.get("RAC_HIP_LIB", "librac_hip.so (shipped)")
    for rank, bad, lq in res:
        print(f"{lib}: {procs} processes, rank {rank}: {bad} of {launches} launches differ from the first; differing "
              f"elements by lane quarter [0-15 16-31 32-47 48-63] = {lq}", flush=True)


if __name__ == "__main__":
    main()

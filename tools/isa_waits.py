"""Static check of the compiler's schedule of the MFMA kernels: how far ahead of its consumer is a load issued?

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Xclang -target-feature -Xclang -packed-fp32-ops \
        --cuda-device-only -S robot_aware_control_amd/csrc/rac_split16.hip -o /tmp/split16.s
  python tools/isa_waits.py /tmp/split16.s [kernel-name-substring]

For every `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` of a kernel's straight-line regions it counts the MFMAs issued between the
awaited load (the N+1-th most recent one) and the wait.  A histogram piled up at 0-2 MFMAs means the scheduler has sunk
the loads to their consumers (an L2 / LDS round trip is then waited out in place): that is how the rows kernel's weight
requests and the persistent kernel's fragment reads were found.  Loop back-edges are not followed, so
kernels whose loads cross an iteration (the tile kernel, the weight-gradient kernels) read as "close" here without being so:
use it on the unrolled kernels, and look at the listing (`--dump`) for the rest."""
import collections
import re
import sys


def kernels(lines):
    cur, body = None, []
    for l in lines:
        m = re.match(r"^(_ZN3rac\w+):", l)
        if m:
            cur, body = m.group(1), []
        elif cur is not None:
            if l.startswith(".Lfunc_end"):
                yield cur, body
                cur = None
            else:
                body.append(l)


def analyse(body):
    vm, lds, mf = [], [], 0
    hist = {"vmcnt": collections.Counter(), "lgkmcnt": collections.Counter()}
    idx = [i for i, l in enumerate(body) if l.strip().startswith("v_mfma")]
    if not idx:
        return 0, hist
    body = body[max(0, idx[0] - 200):idx[-1] + 1]  # the MFMA region (and the loads issued just ahead of it)
    for l in body:
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        if t.startswith("v_mfma"):
            mf += 1
        elif t.startswith(("buffer_load", "global_load")):
            vm.append(mf)
        elif t.startswith(("ds_", "s_load", "s_buffer_load")):  # everything lgkmcnt counts (LDS both ways, scalar loads)
            lds.append(mf)
        elif t.startswith("s_waitcnt") and mf:
            for name, issued in (("vmcnt", vm), ("lgkmcnt", lds)):
                m = re.search(name + r"\((\d+)\)", t)
                if m and len(issued) > int(m.group(1)):
                    d = mf - issued[-1 - int(m.group(1))]
                    hist[name][min(d, 32)] += 1
    return mf, hist


def main():
    src = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    dump = "--dump" in sys.argv
    for name, body in kernels(src):
        if pat not in name:
            continue
        mf, hist = analyse(body)
        if not mf:
            continue
        print(f"{name}: {mf} MFMAs")
        for k in ("vmcnt", "lgkmcnt"):
            h = hist[k]
            tot = sum(h.values())
            if tot:
                close = sum(v for d, v in h.items() if d <= 2)
                print(f"  {k}: {tot} waits, {close} with <= 2 MFMAs since the awaited load; histogram "
                      + " ".join(f"{d}{'+' if d == 32 else ''}:{h[d]}" for d in sorted(h)))
        if dump:
            n = 0
            for l in body:
                t = l.strip()
                if t.startswith("v_mfma"):
                    n += 1
                    continue
                if re.match(r"(buffer_load|global_load|ds_read|ds_write|s_waitcnt|s_barrier|s_cbranch|s_branch|\.LBB)", t):
                    if n:
                        print(f"        MFMA x{n}")
                        n = 0
                    print("    " + t[:90])


if __name__ == "__main__":
    main()

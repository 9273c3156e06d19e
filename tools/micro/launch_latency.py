import torch, time
x = torch.zeros(64, device="cuda")
y = torch.zeros(1 << 22, device="cuda")  # 16 MB
for name, t, n in (("tiny add_", x, 2000), ("16 MB add_", y, 2000)):
    for _ in range(50): t.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): t.add_(1.0)
    e1.record(); host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: GPU {e0.elapsed_time(e1) / n * 1e3:.2f} us per launch, host enqueue {host / n * 1e6:.2f} us per launch")
# graph replay of the tiny chain
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(500): x.add_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"graph of 500 tiny add_: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per kernel")

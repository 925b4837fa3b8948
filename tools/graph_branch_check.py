"""Developer check: do two branches of a captured hipGraph overlap on this ROCm?  main = few large kernels,
side = many tiny kernels; compares replay time of (fork/join) vs (all on one stream)."""
import torch, time
dev = "cuda"
a = torch.randn(4096, 4096, device=dev)
small = [torch.randn(1024, device=dev) for _ in range(64)]
def big():
    x = a
    for _ in range(6):
        x = x @ a * 1e-3
    return x
def tiny():
    for t in small:
        t.mul_(1.0001)
def run(mode):
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    with torch.cuda.stream(s):
        big(); tiny()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            if mode == "fork":
                ev = torch.cuda.Event(); ev.record(s)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    tiny()
                big()
                s.wait_stream(side)
            elif mode == "fork_late":
                a.add_(0.0)
                ev = torch.cuda.Event(); ev.record(s)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    tiny()
                big()
                s.wait_stream(side)
            else:
                tiny(); big()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3
for m in ("serial", "fork", "fork_late", "serial"):
    print(m, "%.3f ms" % run(m))

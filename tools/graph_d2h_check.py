"""Developer check: does a device-to-host copy between two replays of a captured hipGraph break the next replay?
Variants: pure torch graph; graph with a side-stream fork/join; graph containing a memset node."""
import sys, torch
variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
s = torch.cuda.Stream()
side = torch.cuda.Stream()
a = torch.randn(1 << 20, device="cuda")
b = torch.zeros_like(a)
def body():
    global b
    x = a * 2.0 + 1.0
    if variant == "fork":
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        with torch.cuda.stream(side):
            y = (a * 3.0).sum()
        torch.cuda.current_stream().wait_stream(side)
        x = x + y
    if variant == "memset":
        z = torch.zeros(1 << 16, device="cuda", dtype=torch.uint8)  # hipMemsetAsync node
        x = x + z.float().sum()
    b.copy_(x)
with torch.cuda.stream(s):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
print("replays ok", float(b[0]))
print("d2h", torch.zeros(4, device="cuda").cpu().sum().item())
with torch.cuda.stream(s):
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
print("after d2h ok", float(b[0]))

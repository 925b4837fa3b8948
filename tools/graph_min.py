import sys, faulthandler
faulthandler.enable()
import torch
p = torch.nn.Parameter(torch.randn(1000, 3, device="cuda"))
mode = sys.argv[1]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    (p * 2).sum().backward()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if mode == "backward":
        (p * 2).sum().backward()
    else:
        gr, = torch.autograd.grad((p * 2).sum(), p)
        out = gr * 1.0
g.replay()
torch.cuda.synchronize()
print("OK", mode)

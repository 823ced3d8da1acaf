"""torch.optim.Adam fused=True against the default (foreach) capturable path, eager and recorded into a graph, with gradients that are
views of one slab (what autograd.GraphFunction returns) and with owned gradients: the parameters after three steps.  Measured:
identical across capture / views, 2.4e-7 between fused and foreach on unit-scale parameters."""
import torch
dev = torch.device("cuda:0")
shapes = [(100, 32), (100,), (32, 7), (32,), (1, 100), (1,), (32, 32), (5, 3)]
def run(fused, capture, views):
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    kw = dict(fused=True) if fused else {}
    opt = torch.optim.Adam(ps, lr=1e-3, capturable=True, **kw)
    n = sum(p.numel() for p in ps)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    static = torch.randn(n, device=dev, generator=gen)
    def step():
        opt.zero_grad()
        flat = static * 1.0 if views else None
        off = 0
        for p in ps:
            c = p.numel()
            p.grad = (flat[off:off + c].view(p.shape) if views else static[off:off + c].view(p.shape).clone())
            off += c
        opt.step()
    if capture:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for i in range(2):
            static.copy_(torch.randn(n, device=dev, generator=gen)); g.replay()
    else:
        step()
        for i in range(2):
            static.copy_(torch.randn(n, device=dev, generator=gen)); step()
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in ps]).double(), [float(opt.state[p]["step"]) for p in ps][:2]
ref, st = run(False, False, False)
for fused in (False, True):
    for capture in (False, True):
        for views in (False, True):
            out, steps = run(fused, capture, views)
            print("fused", fused, "capture", capture, "views", views, "max diff vs foreach eager", float((out - ref).abs().max()), "steps", steps)

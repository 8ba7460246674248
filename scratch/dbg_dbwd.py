import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gan-pytorch_amd"))
import torch, torch.nn.functional as F
from vegans import _hip
from vegans.engine import ops
gpu = torch.device("cuda:0")
def err(a, b):
    if a is None or b is None:
        return float("nan") if (a is None) != (b is None) else 0.0
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return (a-b).abs().max().item() / max(b.abs().max().item(), 1e-9)
g = torch.Generator().manual_seed(0)
# 1) BN alone: y = bn(x); gx = grad(y·c, x); L = sum(gx^2 * d)
x = torch.randn(6, 8, 5, 5, generator=g); gam = torch.rand(8, generator=g)+0.5; bet = torch.randn(8, generator=g)
c = torch.randn(6, 8, 5, 5, generator=g); d = torch.randn(6, 8, 5, 5, generator=g)
def run(dev, bnf, sq):
    xx = x.detach().clone().to(dev).requires_grad_(True); gg = gam.detach().clone().to(dev).requires_grad_(True); bb = bet.detach().clone().to(dev).requires_grad_(True)
    cc = c.detach().clone().to(dev).requires_grad_(True)
    y = bnf(xx, gg, bb)
    (gx,) = torch.autograd.grad(y, xx, cc, create_graph=True)
    L = sq(gx, d.to(dev))
    L.backward()
    return gx, xx.grad, gg.grad, cc.grad
ref = run("cpu", lambda a, b, c_: F.batch_norm(a, None, None, b, c_, True, 0.1, 1e-5), lambda gx, dd: (gx*dd).sum())
class DotFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(b); return (a*b).sum()
    @staticmethod
    def backward(ctx, go):
        (b,) = ctx.saved_tensors; return go*b, None
hip = run(gpu, lambda a, b, c_: ops.batch_norm(a, b, c_, None, None, True, 0.1, 1e-5), lambda gx, dd: DotFn.apply(gx, ops.to_nhwc(dd)))
print("BN: gx %.2e  d_x %.2e  d_gamma %.2e  d_gy %.2e" % tuple(err(a, b) for a, b in zip(hip, ref)))

# 2) conv alone double backward: y = conv(x,w); gx = grad(y·c, x) ; L = sum(gx*d) -> dL/dw, dL/dc
x = torch.randn(4, 8, 8, 8, generator=g); w = torch.randn(16, 8, 4, 4, generator=g)*0.1
c = torch.randn(4, 16, 4, 4, generator=g); d = torch.randn(4, 8, 8, 8, generator=g)
def runc(dev, convf, sq):
    xx = x.detach().clone().to(dev).requires_grad_(True); ww = w.detach().clone().to(dev).requires_grad_(True); cc = c.detach().clone().to(dev).requires_grad_(True)
    y = convf(xx, ww)
    (gx,) = torch.autograd.grad(y, xx, cc, create_graph=True)
    L = sq(gx, d.to(dev)); L.backward()
    return gx, ww.grad, cc.grad
ref = runc("cpu", lambda a, b: F.conv2d(a, b, None, 2, 1), lambda gx, dd: (gx*dd).sum())
hip = runc(gpu, lambda a, b: ops.conv2d(a, b, None, (2,2), (1,1)), lambda gx, dd: DotFn.apply(gx, ops.to_nhwc(dd)))
print("conv: gx %.2e d_w %.2e d_c %.2e" % tuple(err(a, b) for a, b in zip(hip, ref)))

# 3) conv -> lrelu -> conv, penalty
w2 = torch.randn(1, 16, 4, 4, generator=g)*0.1
def runp(dev, f, pen):
    xx = x.detach().clone().to(dev).requires_grad_(True); ww = w.detach().clone().to(dev).requires_grad_(True); ww2 = w2.detach().clone().to(dev).requires_grad_(True)
    out = f(xx, ww, ww2)
    ones = torch.ones_like(out)
    (gx,) = torch.autograd.grad(out, xx, ones, create_graph=True)
    P = pen(gx); P.backward()
    return gx, P, ww.grad, ww2.grad
ref = runp("cpu", lambda a, b, c_: F.conv2d(F.leaky_relu(F.conv2d(a, b, None, 2, 1), 0.2), c_).flatten(1),
           lambda gx: ((gx.reshape(4, -1).norm(2, dim=1)-1)**2).mean())
hip = runp(gpu, lambda a, b, c_: ops.flatten(ops.conv2d(ops.activation(ops.conv2d(a, b, None, (2,2), (1,1)), _hip.ACT_LRELU, 0.2), c_, None, (1,1), (0,0))),
           ops.gradient_penalty)
print("conv-lrelu-conv GP: gx %.2e P %.2e d_w1 %.2e d_w2 %.2e" % tuple(err(a, b) for a, b in zip(hip, ref)))

import torch, sys
sys.path.insert(0, '.')
import tfep_amd.torch_ops
g = torch.Generator(device='cuda').manual_seed(0)
B, D = 7, 6
x = torch.randn(B, D, device='cuda', generator=g, requires_grad=True)
p = torch.randn(B, 2*D, device='cuda', generator=g, requires_grad=True)
y, l = torch.ops.tfep.affine_forward(x, p)
gx, gp = torch.autograd.grad([y.sum() + l.sum()], [x, p])
x64, p64 = x.detach().double().requires_grad_(True), p.detach().double().requires_grad_(True)
y64 = x64 * torch.exp(p64[:, D:]) + p64[:, :D]
l64 = p64[:, D:].sum(1)
ex, ep = torch.autograd.grad([y64.sum() + l64.sum()], [x64, p64])
print('eager gx err', (gx.double()-ex).abs().max().item(), 'gp err', (gp.double()-ep).abs().max().item())
# only y
y, l = torch.ops.tfep.affine_forward(x, p)
gx2, gp2 = torch.autograd.grad([y.sum()], [x, p])
ex2, ep2 = torch.autograd.grad([(x64 * torch.exp(p64[:, D:]) + p64[:, :D]).sum()], [x64, p64])
print('only-y gx err', (gx2.double()-ex2).abs().max().item(), 'gp err', (gp2.double()-ep2).abs().max().item())
y, l = torch.ops.tfep.affine_forward(x, p)
gx3, gp3 = torch.autograd.grad([l.sum()], [x, p], allow_unused=True)
print('only-l', None if gx3 is None else gx3.abs().max().item(), (gp3.double() - torch.cat([torch.zeros(B,D), torch.ones(B,D)],1).cuda().double()).abs().max().item())
from torch._functorch.aot_autograd import aot_function
def f(x, p):
    y, l = torch.ops.tfep.affine_forward(x, p)
    return y, l
from functorch.compile import nop
af = aot_function(f, nop)
y, l = af(x, p)
gx4, gp4 = torch.autograd.grad([y.sum() + l.sum()], [x, p])
print('aot gx err', (gx4.double()-ex).abs().max().item(), 'gp err', (gp4.double()-ep).abs().max().item())

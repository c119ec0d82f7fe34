"""Host-side profile of a cfg1-sized training step (2-layer MAF + affine, D = 66, B = 1024): where do 3 ms go?"""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.loss import BoltzmannKLDivLoss
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
dev = torch.device('cuda:0')
D, B = 66, 1024
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), initialize_identity=False) for o in ('ascending', 'descending')])
x = torch.randn(B, D, device=dev)
opt = torch.optim.SGD(flow.parameters(), lr=1e-7)
loss_fn = BoltzmannKLDivLoss()


def step():
    opt.zero_grad(set_to_none=True)
    y, l = flow(x)
    loss_fn((y ** 2).sum(dim=1), l).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print('ms per step', (time.perf_counter() - t0) / 50 * 1e3)
# GPU time only: events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)

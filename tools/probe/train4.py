"""Training step of the whole 4-layer cfg2 flow on a cfg3-sized shard (8192 rows): time and peak memory."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.loss import BoltzmannKLDivLoss
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer

D, B = 3000, int(os.environ.get('TRAIN_BATCH', 8192))
dev = torch.device('cuda:0')
torch.manual_seed(0)
with torch.device(dev):
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                initialize_identity=False) for o in ('ascending', 'descending', 'ascending', 'descending')])
x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
c = torch.rand(D, device=dev) * 0.3
opt = torch.optim.SGD(flow.parameters(), lr=1e-6)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    y, ldj = flow(x)
    loss = BoltzmannKLDivLoss()((c * y.clamp(-50, 50) ** 2).sum(dim=1), ldj)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(f'step {it}: {1e3 * (time.perf_counter() - t0):.1f} ms  loss {float(loss):.4f}  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
with torch.no_grad():
    t0 = time.perf_counter()
    xi, _ = flow.inverse(y.detach())
    torch.cuda.synchronize()
    print(f'inverse after training: {1e3 * (time.perf_counter() - t0):.1f} ms  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)

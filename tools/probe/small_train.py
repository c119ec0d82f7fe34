"""Training step of small flows: keep-activations forward (un-fused kernels) against fused forward + recompute."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.loss import BoltzmannKLDivLoss
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow, _backward as bw
from tfep_amd.nn.transformers import NeuralSplineTransformer
dev = torch.device('cuda:0')
for D, B, spline in ((66, 1024, False), (66, 1024, True), (300, 4096, True), (1000, 4096, True), (1000, 16384, True)):
    torch.manual_seed(0)
    with torch.device(dev):
        flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=(NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8)
                                                                     if spline else None), initialize_identity=False)
                                for o in ('ascending', 'descending')])
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    opt = torch.optim.SGD(flow.parameters(), lr=1e-7)

    def step():
        opt.zero_grad(set_to_none=True)
        y, l = flow(x)
        BoltzmannKLDivLoss()((y ** 2).sum(dim=1), l).backward()
        opt.step()
    res = {}
    for name, save in (('keep', bw._SAVE_BYTES or (24 << 30)), ('recompute', 0)):
        bw._SAVE_BYTES = save
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 10 * 1e3
    bw._SAVE_BYTES = 24 << 30
    n_out = flow[0]._conditioner._linears()[-1].out_features
    print(f'D={D} B={B} spline={spline}: theta {B * n_out * 4 / 2**20:.1f} MiB  keep {res["keep"]:.2f} ms  recompute {res["recompute"]:.2f} ms', flush=True)

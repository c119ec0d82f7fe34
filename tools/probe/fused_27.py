"""One cfg2-sized layer with an 8-bin spline whose bounds are learnable (26 / 27 parameters per feature): fused vs un-fused."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer
D, B = 3000, 32768
dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
for ll, lu in ((True, False), (True, True)):
    with torch.device(dev):
        layer = MAF(generate_degrees(D, 'ascending'), initialize_identity=False,
                    transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8, learn_lower_bound=ll, learn_upper_bound=lu))
    res = {'P': layer._transformer.n_parameters_per_feature}
    outs = {}
    with torch.no_grad():
        for fused in (True, False):
            layer.fused = fused
            outs[fused] = layer(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                layer(x)
            torch.cuda.synchronize()
            res['fused_ms' if fused else 'unfused_ms'] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    res['max_abs_dy'] = float((outs[True][0] - outs[False][0]).abs().max())
    res['max_abs_dldj'] = float((outs[True][1] - outs[False][1]).abs().max())
    print(json.dumps(res), flush=True)
    del layer, outs

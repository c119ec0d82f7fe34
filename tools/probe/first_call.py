"""Where does the first call of each path spend its host time (plan building)?  cProfile of one cfg2 layer."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF
from tfep_amd.nn.transformers import NeuralSplineTransformer
D, B = 3000, 4096
t0 = time.perf_counter()
maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
          initialize_identity=False)
print('construct on CPU s', round(time.perf_counter() - t0, 2))
t0 = time.perf_counter(); maf = maf.cuda(); torch.cuda.synchronize(); print('to cuda s', round(time.perf_counter() - t0, 2))
x = torch.randn(B, D, device='cuda').clamp_(-4.9, 4.9)
for name, fn in (('forward', lambda: maf(x)), ('inverse', lambda: maf.inverse(x))):
    pr = cProfile.Profile()
    with torch.no_grad():
        t0 = time.perf_counter(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
        t1 = time.perf_counter(); fn(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{name}: first call {t1 - t0:.2f} s, second {t2 - t1:.3f} s')
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)

"""Blocked inverse of one cfg2 layer at B rows: the whole batch against independent row chunks replayed as HIP graphs on
separate streams (one chunk's block kernel beside another chunk's block GEMMs).  INV_BATCH (default 8192)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd.graphs import GraphedFlow
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import NeuralSplineTransformer

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(os.environ.get('INV_BATCH', 8192))
D = 3000
with torch.device(dev):
    flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                              transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                              initialize_identity=False))
y = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)


def clock(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    print(f'B = {B}: eager whole batch {clock(lambda: flow.inverse(y)):.1f} ms', flush=True)
    x_ref, l_ref = flow.inverse(y)
    g = GraphedFlow(flow, B, D, inverse=True, warmup=1)
    print(f'graph replay, whole batch {clock(lambda: g(y)):.1f} ms', flush=True)
    del g
    for n_chunks in (2, 4):
        rows = B // n_chunks
        graphs = [GraphedFlow(flow, rows, D, inverse=True, warmup=1) for _ in range(n_chunks)]
        streams = [torch.cuda.Stream(dev) for _ in range(n_chunks)]
        outs = [None] * n_chunks

        def run():
            cur = torch.cuda.current_stream(dev)
            for i, (gr, st) in enumerate(zip(graphs, streams)):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs[i] = gr(y[i * rows:(i + 1) * rows])
            for st in streams:
                cur.wait_stream(st)
        ms = clock(run)
        x = torch.cat([o[0] for o in outs])
        print(f'{n_chunks} chunks of {rows} rows on {n_chunks} streams: {ms:.1f} ms; max |x - x_whole| {float((x - x_ref).abs().max()):.2e}', flush=True)
        ms1 = clock(lambda: graphs[0](y[:rows]))
        print(f'   one chunk of {rows} rows alone: {ms1:.1f} ms', flush=True)
        del graphs

import sys, time, torch
sys.path.insert(0, '.')
import bench
flow = bench.build_flow(3000, 1, 8, 'cuda')
layer = flow[0]
made = layer._conditioner
x = torch.randn(256, 3000, device='cuda')
with torch.no_grad():
    flow(x)
    plan = made.plan(x.device)
    fp = layer._fused_plan(x.device, 1, layer._tables(x.device))
    lins = made._linears()
    for li, lin in enumerate(lins):
        kw = dict(row_of_out=fp['row_of_out'], n_rows=fp['n_rows']) if li == 2 else {}
        made._pack_layer_split(plan, li, lin, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            made._pack_layer_split(plan, li, lin, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        n, k = lin.mask.shape
        print(f'layer {li} ({n} x {k}): {dt * 1e3:.3f} ms per pack; v read {n * k * 4 / dt / 1e12:.2f} TB/s')
    # the output layer without the weight-norm pass (g = None): what the row-norm phase costs
    from tfep_amd import ops
    lin = lins[2]
    key = [k for k in plan if isinstance(k, tuple) and k[0] == 'ws' and k[1] == 2][0]
    buf = plan[key]
    cut = made._mask_prefix_cuts(plan, 2, lin)
    for gsel, name in ((lin.weight_g.detach(), 'with norm'), (None, 'no norm')):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.masked_weight_prepare_split(lin.weight_v.detach(), gsel, lin.mask, fp['row_of_out'], plan['in_of_col'][2], buf[0], buf[1], col_cut=cut)
        torch.cuda.synchronize()
        print(name, (time.perf_counter() - t0) / 10 * 1e3, 'ms')

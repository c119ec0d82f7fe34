"""The initial-value steppers of ``tfep_amd/nn/flows/_odeint.py`` on CPU tensors (host logic: no kernel involved).

torchdiffeq is absent, so these are property tests -- "ODE-solve parity unpinned" (SURVEY.md 8c): convergence order of
every tableau, and the failure modes the adaptive pairs must stop on instead of spinning (torchdiffeq asserts on a
non-finite state and on 'underflow in dt'; the reference's loss has ``ignore_nan``, so NaN samples are an expected input).
"""
import math

import pytest
import torch

from tfep_amd.nn.flows import _odeint


def axpy(x, terms):
    out = x.clone()
    for a, v in terms:
        out = out + a * v
    return out


def decay(t, state):                      # y' = -y + sin(t), smooth and non-autonomous
    return (-state[0] + math.sin(t),)


def exact_decay(t, y0):
    # y = c e^{-t} + (sin t - cos t) / 2
    c = y0 + 0.5
    return c * math.exp(-t) + 0.5 * (math.sin(t) - math.cos(t))


@pytest.mark.parametrize('method,order', [('euler', 1), ('midpoint', 2), ('heun3', 3), ('rk4', 4)])
def test_fixed_grid_convergence_order(method, order):
    y0 = torch.tensor([1.0, -0.5, 2.0], dtype=torch.float64)
    errs = []
    for n in (8, 16):
        y, = _odeint.odeint(decay, (y0,), 0.0, 1.0, method=method, options={'step_size': 1.0 / n}, axpy=axpy)
        errs.append(float((y - torch.tensor([exact_decay(1.0, float(v)) for v in y0])).abs().max()))
    assert abs(math.log2(errs[0] / errs[1]) - order) < 0.35


@pytest.mark.parametrize('method', _odeint.ADAPTIVE)
def test_adaptive_pairs_meet_their_tolerance_and_scale_with_it(method):
    y0 = torch.tensor([1.0, -0.5, 2.0], dtype=torch.float64)
    target = torch.tensor([exact_decay(1.0, float(v)) for v in y0])
    errs = []
    for tol in (1e-4, 1e-7):
        stats = {}
        y, = _odeint.odeint(decay, (y0,), 0.0, 1.0, method=method, rtol=tol, atol=tol, axpy=axpy, stats=stats)
        errs.append(float((y - target).abs().max()))
        assert stats['n_steps'] >= 1 and stats['n_evaluations'] >= stats['n_steps']
    slack = 50 if method in ('dopri5', 'bosh3') else 500          # the 2(1) pairs control a much cruder local estimate
    assert errs[0] < slack * 1e-4 and errs[1] < slack * 1e-7 and errs[1] < errs[0]
    # backwards in time returns to the start
    yb, = _odeint.odeint(decay, (target,), 1.0, 0.0, method=method, rtol=1e-8, atol=1e-8, axpy=axpy)
    assert float((yb - y0).abs().max()) < 1e-5


def test_dopri5_error_weights_sum_to_zero_and_embedded_solution_is_fourth_order():
    order, C, A, Bw, E, fsal = _odeint._TABLEAUS['dopri5']
    assert abs(sum(E)) < 1e-15 and fsal and order == 5
    b_star = [b - e for b, e in zip(Bw, E)]
    # order conditions of the embedded 4th-order solution: sum b* c^k = 1 / (k + 1), k = 0..3
    for k in range(4):
        assert abs(sum(b * c ** k for b, c in zip(b_star, C)) - 1.0 / (k + 1)) < 1e-14
    assert abs(b_star[-1] - 1 / 60) < 1e-15            # Shampine's variant (torchdiffeq's), not Dormand-Prince's 1/40


@pytest.mark.parametrize('method', _odeint.ADAPTIVE)
def test_a_nan_sample_stops_the_adaptive_stepper(method):
    def f(t, state):
        return (-state[0],)
    y0 = torch.ones(4, 3, dtype=torch.float64)
    y0[2, 1] = float('nan')
    with pytest.raises(RuntimeError, match='non-finite'):
        _odeint.odeint(f, (y0,), 0.0, 1.0, method=method, axpy=axpy, options={'max_num_steps': 200})


def test_a_finite_time_blow_up_raises_instead_of_spinning():
    def f(t, state):                       # y' = y^2, y(0) = 1: y = 1 / (1 - t) blows up at t = 1
        return (state[0] ** 2,)
    y0 = torch.ones(2, dtype=torch.float64)
    with pytest.raises(RuntimeError, match='underflow in dt|non-finite|max_num_steps'):
        _odeint.odeint(f, (y0,), 0.0, 2.0, method='dopri5', axpy=axpy)


def test_max_num_steps_has_a_finite_default_and_is_enforced():
    def f(t, state):
        return (-state[0],)
    with pytest.raises(RuntimeError, match='max_num_steps'):
        _odeint.odeint(f, (torch.ones(2, dtype=torch.float64),), 0.0, 1.0, method='adaptive_heun', rtol=1e-12, atol=1e-12,
                       axpy=axpy, options={'max_num_steps': 5})

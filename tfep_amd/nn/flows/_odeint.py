"""Initial-value steppers for the continuous flow, in place of the reference's ``torchdiffeq.odeint`` call
(``tfep/nn/flows/continuous.py:136-169``).

torchdiffeq is an un-vendored optional dependency of the reference (absent in the build container), so
nothing here can be compared with reference outputs -- "ODE-solve parity unpinned".  What is implemented are the
published schemes behind torchdiffeq's method names: the fixed-grid ``euler``, ``midpoint``, ``heun3`` and ``rk4``
(torchdiffeq's ``rk4`` is the 3/8-rule variant) with its ``step_size`` option (equal steps, a shorter last one; ``None`` =
one step over the interval), and the adaptive embedded pairs ``dopri5`` (Dormand-Prince 5(4)), ``bosh3``
(Bogacki-Shampine 3(2)), ``fehlberg2`` and ``adaptive_heun`` with an RMS mixed absolute / relative error norm over the
whole state, the 0.9 safety factor and growth limits [0.2, 10], Hairer's initial-step heuristic and FSAL where the pair
has it.  (Not implemented: ``dopri8``, the Adams multistep methods, ``scipy_solver``.)
They are verified by convergence order, step-size control tests and forward / inverse round trips.

The state is a tuple of tensors; ``axpy(x, terms)`` forms ``x + sum a_k v_k`` (``tfep_ode_axpy`` on the HIP path).
"""
import math


FIXED_GRID = ('euler', 'midpoint', 'heun3', 'rk4')
ADAPTIVE = ('dopri5', 'bosh3', 'fehlberg2', 'adaptive_heun')

# Embedded explicit Runge-Kutta pairs as published: (order of the propagated solution, nodes c, stage coefficients a,
# weights b of the propagated solution, error weights e = b - b_embedded, first-same-as-last).
_TABLEAUS = {
    # Dormand & Prince 1980, 5(4), 7 stages, FSAL.  The error weights are Shampine's variant of the embedded pair
    # (b - b*, b* = (1951/21600, 0, 22642/50085, 451/720, -12231/42400, 649/6300, 1/60)): the one torchdiffeq's `dopri5`
    # tabulates as `c_error` (restated from its published tableau; torchdiffeq itself is absent, so the accepted /
    # rejected step sequence remains unverified against it -- "ODE-solve parity unpinned").
    'dopri5': (5, (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0),
               ((), (1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9),
                (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
                (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
                (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)),
               (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0),
               (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
                -2187 / 6784 + 12231 / 42400, 11 / 84 - 649 / 6300, -1 / 60), True),
    # Bogacki & Shampine 1989, 3(2), 4 stages, FSAL
    'bosh3': (3, (0.0, 1 / 2, 3 / 4, 1.0),
              ((), (1 / 2,), (0.0, 3 / 4), (2 / 9, 1 / 3, 4 / 9)),
              (2 / 9, 1 / 3, 4 / 9, 0.0),
              (2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8), True),
    # Fehlberg 2(1), 3 stages (the pair torchdiffeq calls fehlberg2)
    'fehlberg2': (2, (0.0, 1 / 2, 1.0),
                  ((), (1 / 2,), (1 / 256, 255 / 256)),
                  (1 / 512, 255 / 256, 1 / 512),
                  (-1 / 512, 0.0, 1 / 512), False),
    # Heun's method with the Euler step embedded, 2(1)
    'adaptive_heun': (2, (0.0, 1.0),
                      ((), (1.0,)),
                      (1 / 2, 1 / 2),
                      (1 / 2 - 1.0, 1 / 2), False),
}

def _combine(axpy, y, ks, coeffs, dt):
    """``y + dt * sum_j coeffs[j] ks[j]`` for every component of the state."""
    return tuple(axpy(y[c], [(dt * a, k[c]) for a, k in zip(coeffs, ks) if a != 0.0]) for c in range(len(y)))


def _fixed_step(f, method, t0, dt, y, axpy):
    if method == 'euler':
        return _combine(axpy, y, [f(t0, y)], (1.0,), dt)
    if method == 'midpoint':
        k1 = f(t0, y)
        k2 = f(t0 + 0.5 * dt, _combine(axpy, y, [k1], (0.5,), dt))
        return _combine(axpy, y, [k2], (1.0,), dt)
    if method == 'heun3':                                           # Heun's third-order method
        k1 = f(t0, y)
        k2 = f(t0 + dt / 3.0, _combine(axpy, y, [k1], (1 / 3,), dt))
        k3 = f(t0 + dt * 2.0 / 3.0, _combine(axpy, y, [k2], (2 / 3,), dt))
        return _combine(axpy, y, [k1, k3], (0.25, 0.75), dt)
    k1 = f(t0, y)                                                   # rk4, 3/8 rule
    k2 = f(t0 + dt / 3.0, _combine(axpy, y, [k1], (1 / 3,), dt))
    k3 = f(t0 + dt * 2.0 / 3.0, _combine(axpy, y, [k1, k2], (-1 / 3, 1.0), dt))
    k4 = f(t0 + dt, _combine(axpy, y, [k1, k2, k3], (1.0, -1.0, 1.0), dt))
    return _combine(axpy, y, [k1, k2, k3, k4], (0.125, 0.375, 0.375, 0.125), dt)


def _rms(parts):
    """sqrt(mean(square)) over all elements of a list of tensors (one host scalar)."""
    total = sum(float((p.double() ** 2).sum()) for p in parts)
    count = sum(p.numel() for p in parts)
    return math.sqrt(total / max(count, 1))


def _scaled(err, y0, y1, rtol, atol):
    if y1 is None:
        return [e / (atol + rtol * a.abs()) for e, a in zip(err, y0)]
    return [e / (atol + rtol * a.abs().maximum(b.abs())) for e, a, b in zip(err, y0, y1)]


def odeint(f, y0, t0, t1, method='dopri5', options=None, rtol=1e-4, atol=1e-4, axpy=None, stats=None):
    """Integrate ``y' = f(t, y)`` from ``t0`` to ``t1`` (either order); returns the state at ``t1``.

    ``options``: ``step_size`` (fixed grid), ``first_step`` / ``max_num_steps`` (adaptive pairs; ``max_num_steps``
    counts accepted + rejected steps; default 2^31 - 1 like torchdiffeq's, which the reference's ContinuousFlow inherits --
    pass a smaller one through ``ContinuousFlow(solver_options={'max_num_steps': n})`` to bound a run: every step costs a
    full set of dynamics evaluations).
    ``stats``: a dict that receives ``n_steps``, ``n_rejected`` and ``n_evaluations``.

    The adaptive pairs stop loudly instead of spinning: a non-finite state or error estimate (a NaN sample in the batch
    poisons the whole-state RMS norm) and a step size that underflows (``t + dt == t``: a finite-time blow-up, a stiff
    problem) both raise ``RuntimeError``, as torchdiffeq's assertions do."""
    options = dict(options or {})
    y = tuple(y0)
    sign = 1.0 if t1 >= t0 else -1.0
    n_eval = [0]

    def fe(t, state):
        n_eval[0] += 1
        return tuple(f(t, state))

    n_steps = n_rej = 0
    if method in FIXED_GRID:
        step = options.pop('step_size', None)
        span = abs(t1 - t0)
        if step is None:
            grid = [t0, t1]
        else:
            if step <= 0:
                raise ValueError('step_size must be positive')
            n = max(1, int(math.ceil(span / step - 1e-9)))
            grid = [t0 + sign * step * k for k in range(n)] + [t1]
        for a, b in zip(grid[:-1], grid[1:]):
            y = _fixed_step(fe, method, a, b - a, y, axpy)
            n_steps += 1
    elif method in ADAPTIVE:
        order, C, A, Bw, E, fsal = _TABLEAUS[method]
        n_stages = len(C)
        max_steps = int(options.pop('max_num_steps', 2 ** 31 - 1))       # torchdiffeq's default, which the reference's ContinuousFlow inherits
        t = t0
        k1 = fe(t, y)
        h = options.pop('first_step', None)
        if h is None:                                               # Hairer, Norsett & Wanner II.4
            d0, d1 = _rms(_scaled(y, y, None, rtol, atol)), _rms(_scaled(k1, y, None, rtol, atol))
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            y_try = _combine(axpy, y, [k1], (1.0,), sign * h0)
            k_try = fe(t + sign * h0, y_try)
            d2 = _rms(_scaled([b - a for a, b in zip(k1, k_try)], y, None, rtol, atol)) / h0
            h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / order)
            h = min(100.0 * h0, h1)
        h = abs(float(h))
        while (t1 - t) * sign > 1e-12 * max(1.0, abs(t1)):
            if n_steps + n_rej >= max_steps:
                raise RuntimeError(f'{method}: max_num_steps exceeded')
            h = min(h, abs(t1 - t))
            dt = sign * h
            if t + dt == t:
                raise RuntimeError(f'{method}: underflow in dt ({dt!r} at t = {t!r}): the step size control cannot meet '
                                   f'rtol = {rtol}, atol = {atol} (a finite-time blow-up or a stiff problem)')
            ks = [k1]
            for s_ in range(1, n_stages):
                ks.append(fe(t + C[s_] * dt, _combine(axpy, y, ks, A[s_], dt)))
            # (FSAL pairs: the argument of the last stage IS the propagated solution)
            y_new = _combine(axpy, y, ks, Bw, dt)
            err = [sum(dt * e * k[c] for e, k in zip(E, ks) if e != 0.0) for c in range(len(y))]
            ratio = _rms(_scaled(err, y, y_new, rtol, atol))
            if not math.isfinite(ratio):
                raise RuntimeError(f'{method}: non-finite state or error estimate at t = {t!r} (step {dt!r}); the error '
                                   'norm runs over the whole batch, so a single NaN / inf sample stops the integration '
                                   '-- remove such samples before the flow (the loss has ignore_nan for its own inputs)')
            if ratio <= 1.0:
                t, y = t + dt, y_new
                k1 = ks[-1] if fsal else fe(t, y)
                n_steps += 1
            else:
                n_rej += 1
            if ratio == 0.0:
                factor = 10.0
            else:
                factor = min(10.0, max(0.9 * ratio ** (-1.0 / order), 1.0 if ratio < 1.0 else 0.2))
            h *= factor
    else:
        raise ValueError(f"solver must be one of {FIXED_GRID + ADAPTIVE} (torchdiffeq's other methods are not implemented)")
    if options:
        raise ValueError(f'unknown solver options {sorted(options)}')
    if stats is not None:
        stats.update(n_steps=n_steps, n_rejected=n_rej, n_evaluations=n_eval[0])
    return y

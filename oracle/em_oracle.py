"""
ORACLE -- test infrastructure, not product code.

CPU restatement (numpy) of the reference's EM core, used only as the checker in
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
mixemt_amd/ imports this package.

Follows, function by function:
    init_props   /root/reference/mixemt/em.py:23-36
    converged    /root/reference/mixemt/em.py:39-54
    em_step      /root/reference/mixemt/em.py:57-91
    run_em       /root/reference/mixemt/em.py:94-165
and restates the third-party arithmetic that path calls:
    scipy.special.logsumexp -- scipy is UNPINNED in the reference
    (setup.py:19 lists bare 'scipy'); the algorithm restated is the one in
    scipy 1.15.3 (scipy/special/_logsumexp.py:192-247, real-valued branch):
    weights of zero -> -inf; a_max = max, every tie with the max is pulled out
    of the sum; m = (weighted) number of max elements; s = sum(b*exp(a-shift))/m
    over the rest; result = log1p(s) + log(m) + a_max; a non-finite a_max is
    shifted by 0.

Pinning: `logsumexp` here is checked bit-for-bit against scipy's in
tests/test_oracle.py, and em_step/run_em against golden vectors produced by
importing the reference itself (tools/gen_golden.py -> tests/golden/*.npz).
"""

import sys

import numpy


def logsumexp(a, axis=None, b=None):
    """scipy.special.logsumexp (1.15.3), real floating inputs, no keepdims."""
    a = numpy.asarray(a)
    if b is not None:
        b = numpy.asarray(b)
        dt = numpy.result_type(a.dtype, b.dtype, numpy.float64)
        a, b = numpy.broadcast_arrays(a, b)
        a = numpy.array(a, dtype=dt)            # writeable copies
        b = numpy.array(b, dtype=dt)
    else:
        a = numpy.array(a, dtype=numpy.result_type(a.dtype, numpy.float64))
    if a.ndim == 0:
        a = a.reshape(1)
        b = b.reshape(1) if b is not None else None
    ax = tuple(range(a.ndim)) if axis is None else axis
    if a.size == 0:
        shape = list(a.shape)
        for i in (ax if isinstance(ax, tuple) else (ax,)):
            shape[i] = 1
        out = numpy.full(tuple(shape), -numpy.inf, dtype=a.dtype)
        out = numpy.squeeze(out, axis=ax)
        return out[()] if out.ndim == 0 else out

    with numpy.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if b is not None:
            a[b == 0] = -numpy.inf
        a_max = numpy.max(a, axis=ax, keepdims=True)
        at_max = a == a_max
        a[at_max] = -numpy.inf
        at_max_f = at_max.astype(a.dtype)
        if b is None:
            m = numpy.sum(at_max_f, axis=ax, keepdims=True, dtype=a.dtype)
        else:
            m = numpy.sum(b * at_max_f, axis=ax, keepdims=True, dtype=a.dtype)
        shift = numpy.where(numpy.isfinite(a_max), a_max,
                            numpy.asarray(0, dtype=a_max.dtype))
        e = numpy.exp(a - shift)
        if b is not None:
            e = b * e
        s = numpy.sum(e, axis=ax, keepdims=True, dtype=e.dtype)
        s = numpy.where(s == 0, s, s / m)

        def _sign(x):
            return x / numpy.where(x == 0, numpy.asarray(1, dtype=x.dtype),
                                   numpy.abs(x))
        sgn = _sign(s + 1) * _sign(m)
        s = numpy.where(s < -1, -s - 2, s)
        m = numpy.abs(m)
        out = numpy.log1p(s) + numpy.log(m) + a_max
        out[sgn < 0] = numpy.nan
    out = numpy.squeeze(out, axis=ax)
    return out[()] if out.ndim == 0 else out


def init_props(nhaps, alpha=1.0):
    """em.py:23-36 -- draws from numpy's process-global legacy RNG."""
    if alpha == float("inf"):
        return numpy.array([1.0 / nhaps] * nhaps)
    return numpy.random.dirichlet([alpha] * nhaps)


def converged(prop, last_prop, tolerance=0.0001):
    """em.py:39-54 -- L1 distance of the exp'd log-proportions."""
    return numpy.sum(numpy.abs(numpy.exp(prop)
                               - numpy.exp(last_prop))) < tolerance


def em_step(read_hap_mat, weights, ln_props, read_mix_mat):
    """em.py:57-91 -- one E+M step; writes and returns read_mix_mat."""
    numpy.add(ln_props, read_hap_mat, read_mix_mat)
    numpy.subtract(read_mix_mat,
                   logsumexp(read_mix_mat, axis=1).reshape((-1, 1)),
                   read_mix_mat)
    new_props = logsumexp(read_mix_mat, axis=0, b=weights.reshape((-1, 1)))
    new_props -= logsumexp(new_props)
    return read_mix_mat, new_props


def _one_run(read_hap_mat, weights, ln_init, post_buf, max_iter, tol, verbose):
    """
    One EM run from log-proportions `ln_init` (em.py:126-143 without the buffer
    swapping).  Returns (theta_next, posterior, iterations) where the posterior
    is the E-step under the proportions ONE step before theta_next -- the
    reference returns exactly that pair, both on convergence and when max_iter
    is exhausted (the for-else at em.py:141-143 swaps the pair back).
    """
    theta, theta_next, n_iter = ln_init, ln_init, 0
    while n_iter < max_iter:
        n_iter += 1
        if verbose and n_iter % 10 == 0:
            sys.stderr.write(".")
        post_buf, theta_next = em_step(read_hap_mat, weights, theta, post_buf)
        if converged(theta, theta_next, tol):
            if verbose:
                sys.stderr.write("\nConverged! (%d)\n" % n_iter)
            break
        theta = theta_next
    return theta_next, post_buf, n_iter


def run_em(read_hap_mat, weights, args, trace=None):
    """
    em.py:94-165: n_multi runs from sequential init draws; the first run's
    results are kept, later runs are folded in as  sum of LOG proportions
    (:155) and logaddexp of posteriors (:156); with n_multi > 1 the sums are
    divided by n / shifted by log n (:158-161) -- a geometric mean that is not
    renormalised; proportions are returned linear, posteriors in log (:163-165).

    `trace`, if a list, receives one dict per run (init draw, iteration count,
    that run's linear proportions): parity observables that the reference
    computes but does not return.
    """
    nhaps = read_hap_mat.shape[1]
    sum_ln_props, acc_post = None, None
    for run in range(args.n_multi):
        if args.verbose:
            sys.stderr.write("Starting EM run %d...\n" % (run + 1))
        init = init_props(nhaps, alpha=args.init_alpha)
        ln_props, post, n_iter = _one_run(
            read_hap_mat, weights, numpy.log(init),
            numpy.empty_like(read_hap_mat), args.max_iter, args.tolerance,
            args.verbose)
        if trace is not None:
            trace.append({"init": init, "iters": n_iter,
                          "props": numpy.exp(ln_props)})
        if run == 0:
            sum_ln_props, acc_post = ln_props, post
        else:
            sum_ln_props += ln_props
            numpy.logaddexp(acc_post, post, acc_post)
    if args.n_multi > 1:
        sum_ln_props /= args.n_multi
        acc_post -= numpy.log(args.n_multi)
    return numpy.exp(sum_ln_props), acc_post

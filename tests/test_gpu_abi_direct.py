"""
The C ABI called directly (ctypes, raw pointers): padded leading dimensions,
NULL weights, sub-streams, and the error contract (negative status + message,
no exception, no crash) -- what a non-Python binder of include/mixemt_hip.h
relies on.
"""
import numpy
import pytest

from oracle import em_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mixemt_amd import _lib
    return _lib.load()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def test_padded_leading_dimensions_and_null_weights(lib):
    """M with ldm > H, P with ldp > H+1, out with ldo > H; w = NULL means weight 1."""
    import torch
    rng = numpy.random.default_rng(3)
    n_rows, n_haps, ldm, ldp, ldo = 90, 700, 704, 712, 710
    host = rng.normal(-20.0, 6.0, size=(n_rows, n_haps))
    wide = torch.full((n_rows, ldm), float("nan"), dtype=torch.float64, device="cuda")
    wide[:, :n_haps] = torch.from_numpy(host).cuda()
    lin = torch.full((n_rows, ldp), float("nan"), dtype=torch.float64, device="cuda")
    rowmax = torch.empty(n_rows, dtype=torch.float64, device="cuda")
    s = _stream()
    assert lib.mxm_linearize(wide.data_ptr(), ldm, n_rows, n_haps, lin.data_ptr(), ldp, rowmax.data_ptr(), s) == 0
    got = lin.cpu().numpy()
    assert numpy.allclose(got[:, :n_haps], numpy.exp(host - host.max(axis=1, keepdims=True)), rtol=1e-15, atol=0)
    assert (got[:, n_haps:] == 0).all()                      # pad columns are zeroed by the call
    props = rng.dirichlet([1.0] * n_haps)
    p_d = torch.from_numpy(props[None, :]).cuda()
    colsum = torch.zeros_like(p_d)
    nbytes = lib.mxm_workspace_bytes(n_rows, n_haps, 1)
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device="cuda")
    lp_d = torch.log(p_d)
    rc = lib.mxm_em_iter(wide.data_ptr(), ldm, lin.data_ptr(), ldp, None, p_d.data_ptr(), lp_d.data_ptr(),
                         n_rows, n_haps, 1, None, colsum.data_ptr(), ws.data_ptr(), nbytes, s)
    assert rc == 0
    mix, new = em_oracle.em_step(host, numpy.ones(n_rows), numpy.log(props), numpy.empty_like(host))
    got = colsum[0].cpu().numpy() * props                     # the call returns the sums without p_h
    assert numpy.allclose(got / got.sum(), numpy.exp(new), rtol=0, atol=1e-13)
    out = torch.full((n_rows, ldo), 7.0, dtype=torch.float64, device="cuda")
    lnp = torch.from_numpy(numpy.log(props)).cuda()
    assert lib.mxm_em_step(wide.data_ptr(), ldm, None, lnp.data_ptr(), n_rows, n_haps, out.data_ptr(), ldo, 0,
                           None, None, 0, s) == 0
    o = out.cpu().numpy()
    assert numpy.allclose(o[:, :n_haps], mix, rtol=0, atol=1e-10) and (o[:, n_haps:] == 7.0).all()


def test_error_contract(lib):
    """Bad arguments: negative status, message through mxm_last_error, nothing launched."""
    import torch
    x = torch.zeros(64, dtype=torch.float64, device="cuda")
    s = _stream()
    assert lib.mxm_em_iter(x.data_ptr(), 8, None, 0, None, x.data_ptr(), x.data_ptr(), 0, 8, 1, None,
                           x.data_ptr(), None, 0, s) < 0
    assert b"mxm_em_iter" in lib.mxm_last_error()
    # odd ldp for the linear matrix
    assert lib.mxm_linearize(x.data_ptr(), 8, 8, 8, x.data_ptr(), 9, x.data_ptr(), s) < 0
    assert b"ldp" in lib.mxm_last_error()
    # workspace too small
    big = torch.zeros((4, 128), dtype=torch.float64, device="cuda")
    assert lib.mxm_em_iter(big.data_ptr(), 128, big.data_ptr(), 128, None, big.data_ptr(), big.data_ptr(), 4, 128,
                           1, None, big.data_ptr(), big.data_ptr(), 16, s) < 0
    assert b"workspace" in lib.mxm_last_error()
    assert lib.mxm_set_batch_tile(7) < 0 and lib.mxm_set_batch_tile(0) < 0 and lib.mxm_set_batch_tile(4) == 0
    # build: lde not a multiple of 8
    assert lib.mxm_build_em_matrix(x.data_ptr(), 7, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                   x.data_ptr(), 1, 5, 3, x.data_ptr(), 5, s) < 0
    assert b"lde" in lib.mxm_last_error()
    torch.cuda.synchronize()                                   # the device is still healthy


def test_em_loop_on_a_side_stream(lib):
    """Every call only enqueues on the stream it is given; mxm_em_loop orders itself after it."""
    import torch
    from mixemt_amd import _lib, em
    rng = numpy.random.default_rng(5)
    mat = rng.normal(-20.0, 6.0, size=(200, 300))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        res = em.run_em_ex(mat, numpy.ones(200), __import__("argparse").Namespace(
            init_alpha=1.0, tolerance=1e-4, max_iter=50, n_multi=2, verbose=False),
            inits=rng.dirichlet([1.0] * 300, size=2))
    side.synchronize()
    for run in range(2):
        theta = numpy.log(res["inits"][run])
        buf = numpy.empty_like(mat)
        for _ in range(res["iters"][run]):
            buf, theta = em_oracle.em_step(mat, numpy.ones(200), theta, buf)
        assert numpy.abs(res["run_props"][run] - numpy.exp(theta)).max() < 1e-11


def test_out_of_device_memory_is_a_value_error():
    """A matrix that cannot fit reports ValueError (what bin/mixemt:325-327 catches), not a torch exception."""
    import torch
    from mixemt_amd._dev import device_empty
    with pytest.raises(ValueError, match="not enough device memory"):
        device_empty((60000000, 5408), torch.float64, torch.device("cuda"), "a 2.6 TB matrix")

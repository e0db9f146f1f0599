import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def co():
    """The C oracle (checker)."""
    from oracle import coracle

    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def pr():
    from oracle import pyref

    return pyref


@pytest.fixture(scope="session")
def ps_api():
    """The product's host mirror; importing it loads libplaysnark_hip.so (fails loudly if absent)."""
    from playsnark_amd import api

    return api


@pytest.fixture(scope="session")
def ctx(ps_api):
    c = ps_api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def off_subgroup(pr):
    """(G1 point, G2 point) on the curve E(Fp) / the twist E'(Fp2) but NOT in the order-r subgroup: the cofactors
    are huge, so the first x with a square right-hand side almost surely gives one.  [r]P is computed as
    (r-1)P + P because the oracle's Mul reduces its scalar mod r, like kyber's."""
    times_r = lambda grp, pt: grp.add(grp.mul(pr.R - 1, pt), pt)
    x = 4
    while True:
        y = pr.fp_sqrt((x * x * x + 4) % pr.P)
        if y is not None and times_r(pr.G1, (x, y)) is not None:
            g1 = (x, y)
            break
        x += 1
    x = 1
    while True:
        X = (x, 1)
        y = pr.f2_sqrt(pr.f2_add(pr.f2_mul(pr.f2_sqr(X), X), (4, 4)))
        if y is not None and times_r(pr.G2, (X, y)) is not None:
            g2 = (X, y)
            break
        x += 1
    assert pr.G1.on_curve(g1) and pr.G2.on_curve(g2)
    return g1, g2


class _HipMem:
    """Raw device buffers for the tests that hand the library a device pointer (ps_scalars_from_device_be32),
    through the HIP runtime the library itself is linked to -- not through torch, whose bundled runtime cannot
    initialise once another copy of libamdhip64 owns the process."""

    def __init__(self):
        import ctypes

        self.C = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.live = []

    def alloc(self, nbytes, data=None):
        C = self.C
        ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(ptr), C.c_size_t(nbytes)) == 0
        self.live.append(ptr)
        if data is None:
            assert self.hip.hipMemset(ptr, 0, C.c_size_t(nbytes)) == 0
        else:
            assert len(data) == nbytes
            assert self.hip.hipMemcpy(ptr, data, C.c_size_t(nbytes), 1) == 0  # hipMemcpyHostToDevice
        assert self.hip.hipDeviceSynchronize() == 0
        return ptr.value

    def free_all(self):
        for p in self.live:
            self.hip.hipFree(p)
        self.live = []


@pytest.fixture
def hipmem(ctx):
    m = _HipMem()
    yield m
    m.free_all()

"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the
same seeded inputs.  Bar: bit-exact frames (the arithmetic is IEEE-identical by construction)."""
import numpy as np
import pytest

from wurblpt_amd import host

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum()) / np.sqrt((b.astype(np.float64) ** 2).sum()))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from wurblpt_amd import device
    assert device.device_count() >= 1
    return device


@pytest.mark.parametrize("op,lo,hi", [(0, -10, 10), (1, -10, 10), (2, -90, 88), (4, -1, 1), (6, -100, 100), (7, 0, 1e6), (8, -3, 3), (9, -100, 100)])
def test_device_arithmetic_is_bit_identical_to_host(dev, oracle, op, lo, hi):
    """sin/cos/exp/asin (wpt_math.h), IEEE division, IEEE sqrt, unfused multiply-add, reciprocal"""
    rng = np.random.RandomState(op)
    a = rng.uniform(lo, hi, 1 << 20).astype(np.float32)
    b = rng.uniform(-7, 7, 1 << 20).astype(np.float32)
    got = dev.selftest_math(op, a, b)
    if op <= 5:
        ref = oracle.math(op, a, b)
    elif op == 6:
        ref = a / b
    elif op == 7:
        ref = np.sqrt(a)
    elif op == 8:
        ref = (a * b).astype(np.float32) + a
    else:
        ref = np.float32(1.0) / a
    assert np.array_equal(got.view(np.uint32), ref.astype(np.float32).view(np.uint32))


def test_device_pow_atan2_bit_identical(dev, oracle):
    rng = np.random.RandomState(5)
    x = rng.uniform(0, 1, 1 << 20).astype(np.float32)
    y = rng.uniform(0, 300, 1 << 20).astype(np.float32)
    assert np.array_equal(dev.selftest_math(3, x, y).view(np.uint32), oracle.math(3, x, y).view(np.uint32))
    a = rng.uniform(-5, 5, 1 << 20).astype(np.float32)
    b = rng.uniform(-5, 5, 1 << 20).astype(np.float32)
    assert np.array_equal(dev.selftest_math(5, a, b).view(np.uint32), oracle.math(5, a, b).view(np.uint32))


@pytest.mark.parametrize("tall,short", [(0, 0), (1, 2)])
def test_cornell_frame_bit_exact(dev, oracle, tall, short):
    """configs 1 and 2 at a size the oracle finishes in seconds"""
    sc = host.cornell(96, 96, tall, short)
    ref, rc = oracle.render(sc, 6)
    ds = dev.DeviceScene(sc)
    got, gc = ds.render(6, with_counters=True)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 1e-4, rel_l2(got, ref)
    nbad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    assert nbad == 0, "%d of %d values differ, rel-L2 %.3g, max abs %.3g" % (nbad, got.size, rel_l2(got, ref), np.abs(got - ref).max())
    assert gc == rc, (gc, rc)

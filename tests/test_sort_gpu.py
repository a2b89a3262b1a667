"""csrc/radix_sort.hip (round 6: the library's own stable LSD radix sort and exclusive scan, in place of rocPRIM's):
against numpy's stable sort / cumsum at sizes around the 2 048-element tiles and the 64-element chunks, every key width,
heavy duplicates (stability is what the deterministic scatters rely on), through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(keys, vals, bits, dev):
    from s4g_release_amd import _cabi
    n = len(keys)
    ki = torch.from_numpy(keys.astype(np.int64)).to(dev).to(torch.int64).to(torch.int32) if False else \
        torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    vi = torch.from_numpy(vals.view(np.int32).copy()).to(dev)
    ko, vo = torch.empty_like(ki), torch.empty_like(vi)
    nb = _cabi.lib().s4g_sort_pairs_workspace_bytes(n)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    rc = _cabi.lib().s4g_sort_pairs_u32(ki.data_ptr(), vi.data_ptr(), n, bits, ko.data_ptr(), vo.data_ptr(),
                                        ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "sort_pairs")
    torch.cuda.synchronize()
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4096 + 17, 100000, 1 << 20, 5242880 + 3])
def test_stable_sort_matches_numpy(dev, n):
    rng = np.random.default_rng(n)
    for bits, hi in ((32, 1 << 32), (26, 1 << 26), (17, 1 << 17), (9, 300), (8, 256), (3, 8), (1, 2)):
        keys = rng.integers(0, hi, size=n, dtype=np.uint64).astype(np.uint32)
        if bits < 32:
            keys &= np.uint32((1 << bits) - 1)
        vals = np.arange(n, dtype=np.uint32)                        # positions: stability is visible in them
        order = np.argsort(keys, kind="stable")
        ko, vo = _sort(keys, vals, bits, dev)
        assert np.array_equal(ko, keys[order]), (n, bits)
        assert np.array_equal(vo, vals[order]), (n, bits)           # equal keys in input order
        if n > (1 << 20):
            break                                                   # (the large size once, at full key width)


def test_sort_ignores_the_bits_above_the_requested_width(dev):
    """LSD on the low `bits` only: the order among keys that agree on those bits is the INPUT order."""
    rng = np.random.default_rng(5)
    n = 30000
    keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    ko, vo = _sort(keys, vals, 12, dev)
    order = np.argsort(keys & np.uint32(0xFFF), kind="stable")
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])
    ko, vo = _sort(keys, vals, 0, dev)                              # no pass at all: a copy
    assert np.array_equal(ko, keys) and np.array_equal(vo, vals)


@pytest.mark.parametrize("n", [1, 255, 256, 2047, 2048, 2049, 70000, (1 << 21) + 5])
def test_exclusive_scan_matches_numpy(dev, n):
    from s4g_release_amd import _cabi
    rng = np.random.default_rng(n)
    x = rng.integers(0, 3, size=n).astype(np.int32)
    d = torch.from_numpy(x).to(dev)
    out = torch.empty_like(d)
    nb = _cabi.lib().s4g_exclusive_scan_workspace_bytes(n)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    rc = _cabi.lib().s4g_exclusive_scan_i32(d.data_ptr(), out.data_ptr(), n, ws.data_ptr(), nb,
                                            torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "exclusive_scan")
    want = np.concatenate([[0], np.cumsum(x.astype(np.int64))[:-1]]).astype(np.int32)
    assert np.array_equal(out.cpu().numpy(), want)
    assert _cabi.lib().s4g_exclusive_scan_i32(d.data_ptr(), d.data_ptr(), n, ws.data_ptr(), nb, None) == _cabi.S4G_EINVAL

"""Tile-book form of the slab conv kernels (csrc/sparse_conv_slab_tb.h): the packed book against a numpy restatement of its
layout, and pvx_sparse_conv_fwd_tb bit-identical to pvx_sparse_conv_fwd (same MFMA order, same epilogue) for every shape that
has the form -- rows in key order (every neighbour inside the window), rows shuffled (most neighbours outside: the far code
and its int32 fallback), live rows below the capacity, strided rulebooks; and against the oracle (bf16 bar 1e-2)."""
import ctypes

import numpy as np
import pytest
import torch

from openpcdet_amd import _lib
from oracle import loader

pytestmark = pytest.mark.gpu

SHAPES = [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)]    # the last two: streaming kernels, spans only


def tile_blocks(book, tiles):
    """the first `tiles` 866 x 16-byte tile blocks of a book buffer as uint32 rows (the schedule area lies behind the blocks)"""
    return book.cpu().numpy()[:tiles * 866 * 16].view(np.uint32).reshape(tiles, 866 * 4)


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def sites(seed, batch, shape, n, sort):
    r = np.random.default_rng(seed)
    c = np.stack([r.integers(0, batch, n), r.integers(0, shape[0], n), r.integers(0, shape[1], n), r.integers(0, shape[2], n)], 1)
    c = np.unique(c, axis=0).astype(np.int32)               # np.unique sorts rows lexicographically = key order
    if not sort:
        r.shuffle(c, axis=0)
    return c


def book_numpy(nbr, n_live, n_in, rcap):
    """the layout documented in sparse_conv_slab_tb.h"""
    K, cap = nbr.shape
    tiles = (n_live + 255) // 256
    meta = np.zeros((tiles, 8), np.uint32)
    ent = np.full((tiles, 27, 256), 0xFFFF, np.uint32)
    for t in range(tiles):
        s0, s1 = t * 256, min(t * 256 + 256, n_live)
        for g in range(3):
            blk = nbr[g * 9:(g + 1) * 9, s0:s1].astype(np.int64)
            blk = np.where(blk >= n_in, -1, blk)
            if (blk >= 0).any():
                lo, hi = blk[blk >= 0].min(), blk[blk >= 0].max()
                span = hi - lo + 1
                cnt = min(span, rcap)
                wlo = lo if span <= rcap else lo + (span - rcap) // 2
                meta[t, 2 * g], meta[t, 2 * g + 1] = wlo, cnt
                meta[t, 6] |= (1 << g) | ((16 << g) if span <= rcap else 0)
                local = blk - wlo
                code = np.where(blk < 0, 0xFFFF, np.where((local >= 0) & (local < cnt), local, 0xFFFE))
                ent[t, g * 9:(g + 1) * 9, :s1 - s0] = code
                # word 7: far steps of the group = max over the tile's four 64-site waves of its (offset, 32-site block) pairs with a far code
                farb = (ent[t, g * 9:(g + 1) * 9] == 0xFFFE).reshape(9, 4, 2, 32).any(-1)        # [offset, wave, block]
                meta[t, 7] |= int(farb.sum((0, 2)).max()) << (5 * g)
    return meta, ent


def run_conv(dev, feat, w, nbr_t, n_live_t, cin, cout, code, book=None, rcap=0, residual=None):
    L = _lib.lib()
    cap = nbr_t.shape[1]
    pw = torch.empty(L.pvx_sparse_conv_weight_bytes(cout, 27, cin, code), dtype=torch.uint8, device=dev)
    _lib.check(L.pvx_sparse_conv_prepare_weight(_lib.ptr(w), cout, 27, cin, code, _lib.ptr(pw), _lib.stream()), "prepare")
    out = torch.full((cap, cout), 7.0, dtype=feat.dtype, device=dev)
    scale = torch.linspace(0.5, 1.5, cout, device=dev)
    shift = torch.linspace(-0.2, 0.2, cout, device=dev)
    bias = torch.linspace(-0.1, 0.1, cout, device=dev)
    _lib.check(L.pvx_sparse_conv_fwd_tb(_lib.ptr(feat), feat.shape[0], cin, _lib.ptr(pw), _lib.ptr(bias), _lib.ptr(nbr_t), 27, cap,
                                        _lib.ptr(n_live_t), cout, code, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), 1,
                                        _lib.ptr(out), _lib.ptr(book), rcap, _lib.stream()), "pvx_sparse_conv_fwd_tb")
    return out, (scale, shift, bias)


@pytest.mark.parametrize("cin,cout", SHAPES)
@pytest.mark.parametrize("sort", [True, False])
def test_tilebook_conv_bit_identical(cuda_device, cin, cout, sort):
    L = _lib.lib()
    dev = cuda_device
    shape, batch = [12, 60, 56], 2
    coords = sites(5 + cin, batch, shape, 30000, sort)
    n = len(coords)
    cap = n + 300                                            # capacity above the live count
    nbr = np.full((27, cap), -1, np.int32)
    nbr[:, :n] = loader.subm_rulebook(coords, shape, (3, 3, 3))
    r = np.random.default_rng(9)
    for dtype, code in ((torch.bfloat16, 2), (torch.float16, 1)):
        rcap = L.pvx_tilebook_rcap(cin, cout, 27, code)
        assert rcap > 0
        feat = torch.from_numpy(r.normal(size=(cap, cin)).astype(np.float32)).to(dev).to(dtype)
        w = torch.from_numpy((r.normal(size=(cout, 27, cin)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev).to(dtype)
        res = torch.from_numpy(r.normal(size=(cap, cout)).astype(np.float32)).to(dev).to(dtype) if cin == cout else None
        nbr_t = torch.from_numpy(nbr).to(dev)
        n_live = torch.tensor([n], dtype=torch.int32, device=dev)
        book = torch.zeros(L.pvx_tilebook_bytes(cap), dtype=torch.uint8, device=dev)
        _lib.check(L.pvx_tilebook_build(_lib.ptr(nbr_t), 27, cap, _lib.ptr(n_live), cap, rcap, _lib.ptr(book), _lib.stream()), "build")
        # the book itself
        meta, ent = book_numpy(nbr, n, cap, rcap)
        raw = tile_blocks(book, meta.shape[0])
        assert np.array_equal(raw[:, :8], meta[:, :8])
        got_ent = raw[:, 8:].copy().view(np.uint16).reshape(-1, 27, 256)
        assert np.array_equal(got_ent, ent.astype(np.uint16))
        far = (ent == 0xFFFE).mean()
        # sorted rows fit the window (the 320-row window of the 128-channel form: all but a few lanes); shuffled rows mostly do not
        assert ((far == 0) if rcap >= 384 else (far < 0.05)) if sort else (far > 0.05), far
        # the conv
        want, (scale, shift, bias) = run_conv(dev, feat, w, nbr_t, n_live, cin, cout, code, residual=res)
        got, _ = run_conv(dev, feat, w, nbr_t, n_live, cin, cout, code, book=book, rcap=rcap, residual=res)
        assert torch.equal(got[:n], want[:n]), (cin, cout, dtype)
        assert torch.equal(got[n:], want[n:])                # rows beyond the live count untouched by both
        if dtype == torch.bfloat16:
            ref = loader.sparse_conv_fwd(feat[:n].float().cpu().numpy(), w.float().cpu().numpy(), bias.cpu().numpy(), nbr[:, :n])
            ref = ref * scale.cpu().numpy() + shift.cpu().numpy()
            if res is not None:
                ref = ref + res[:n].float().cpu().numpy()
            assert rel_l2(got[:n].float().cpu().numpy(), np.maximum(ref, 0)) <= 1e-2
        # a book built for another window size is refused in favour of the plain kernels (still the same result)
        got2, _ = run_conv(dev, feat, w, nbr_t, n_live, cin, cout, code, book=book, rcap=rcap + 1, residual=res)
        assert torch.equal(got2[:n], want[:n])


@pytest.mark.parametrize("n_sites,sort", [(150000, True), (150000, False), (9000, True)])
def test_tile_schedule(cuda_device, n_sites, sort):
    """pvx_tilebook_schedule: every live tile of an XCD's eighth lands in exactly one workgroup's list, in ascending order, the lists
    are balanced by the cost model's own measure, a rebuilt book invalidates the schedule, and the conv that walks the schedule gives
    the bits of the plain kernels (with a live count below the capacity, and a second conv after the live count changed)."""
    L = _lib.lib()
    dev = cuda_device
    shape, batch = [12, 200, 220], 2
    coords = sites(61, batch, shape, n_sites, sort)
    n = len(coords)
    cap = n + 700
    nbr = np.full((27, cap), -1, np.int32)
    nbr[:, :n] = loader.subm_rulebook(coords, shape, (3, 3, 3))
    nbr_t = torch.from_numpy(nbr).to(dev)
    r = np.random.default_rng(8)
    tested = 0
    for cin, cout in SHAPES + [(128, 128)]:
        if L.pvx_tilebook_schedule(None, cap, None, cin, cout, 27, 2, None) != 1:
            continue
        tested += 1
        rcap = L.pvx_tilebook_rcap(cin, cout, 27, 2)
        feat = torch.from_numpy(r.normal(size=(cap, cin)).astype(np.float32)).to(dev).bfloat16()
        w = torch.from_numpy((r.normal(size=(cout, 27, cin)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev).bfloat16()
        res = torch.from_numpy(r.normal(size=(cap, cout)).astype(np.float32)).to(dev).bfloat16()
        book = torch.full((L.pvx_tilebook_bytes(cap),), 0xA5, dtype=torch.uint8, device=dev)       # garbage behind the blocks
        tiles_alloc = (cap + 255) // 256
        for live in (n, n // 3):
            n_live = torch.tensor([live], dtype=torch.int32, device=dev)
            _lib.check(L.pvx_tilebook_build(_lib.ptr(nbr_t), 27, cap, _lib.ptr(n_live), cap, rcap, _lib.ptr(book), _lib.stream()), "build")
            sched = book[tiles_alloc * 866 * 16:].view(torch.int32)
            assert int(sched[0]) == 0                       # the builder invalidates whatever was there
            want, _ = run_conv(dev, feat, w, nbr_t, n_live, cin, cout, 2, residual=res)
            unsched, _ = run_conv(dev, feat, w, nbr_t, n_live, cin, cout, 2, book=book, rcap=rcap, residual=res)
            assert torch.equal(unsched, want)
            _lib.check(L.pvx_tilebook_schedule(_lib.ptr(book), cap, _lib.ptr(n_live), cin, cout, 27, 2, _lib.stream()), "schedule")
            hdr = sched[:16].cpu().numpy()
            ntiles = (live + 255) // 256
            W, R = int(hdr[1]), int(hdr[2])
            assert hdr[0] == 0x5c4ed01e and hdr[3] == ntiles and all(hdr[4 + x] == 0x5c4ed01e for x in range(8)) and 1 <= W <= 64
            counts = sched[16:16 + 512].cpu().numpy().reshape(8, 64)
            lists = sched[16 + 512:16 + 512 + 512 * R].cpu().numpy().reshape(8, 64, R)
            blocks = tile_blocks(book, ntiles)
            meta, steps = blocks[:, 6] | 2, blocks[:, 7]          # flags (group 1 always runs), far steps per group (5 bits each)
            cost = 12 + sum(np.where((meta >> g) & 1, np.where((meta >> (4 + g)) & 1, 40, 48 + 5 * ((steps >> (5 * g)) & 31)), 0)
                            for g in range(3)).astype(np.int64)
            for x in range(8):
                lo, hi = ntiles * x // 8, ntiles * (x + 1) // 8
                got = []
                loads = []
                for wq in range(W):
                    c = int(counts[x, wq])
                    mine = lists[x, wq, :c]
                    assert np.all(np.diff(mine) > 0)        # ascending: neighbouring tiles stay neighbours in time
                    got.extend(mine.tolist())
                    loads.append(int(cost[mine].sum()))
                assert sorted(got) == list(range(lo, hi)), x
                if hi - lo >= 2 * W:
                    assert max(loads) - min(loads) <= 2 * int(cost.max()), (x, loads)      # within two of the heaviest tiles of each other
            got, _ = run_conv(dev, feat, w, nbr_t, n_live, cin, cout, 2, book=book, rcap=rcap, residual=res)
            assert torch.equal(got, want), (cin, cout, live)
        # a schedule made for another live count is ignored (the kernel compares it with its own)
        n_other = torch.tensor([n // 2], dtype=torch.int32, device=dev)
        _lib.check(L.pvx_tilebook_build(_lib.ptr(nbr_t), 27, cap, _lib.ptr(n_other), cap, rcap, _lib.ptr(book), _lib.stream()), "build")
        sched[0] = 0x5c4ed01e                               # pretend the old schedule (made for n // 3 rows) were still valid
        want, _ = run_conv(dev, feat, w, nbr_t, n_other, cin, cout, 2, residual=res)
        got, _ = run_conv(dev, feat, w, nbr_t, n_other, cin, cout, 2, book=book, rcap=rcap, residual=res)
        assert torch.equal(got, want)
    assert tested > 0
    assert L.pvx_tilebook_schedule(None, cap, None, 16, 16, 27, 2, None) == 0 and L.pvx_tilebook_schedule(None, -1, None, 64, 64, 27, 2, None) == -1


def test_tilebook_strided_rulebook_and_empty(cuda_device):
    L = _lib.lib()
    dev = cuda_device
    shape = [9, 40, 44]
    coords = sites(31, 2, shape, 15000, True)
    oc, nbr, oshape = loader.sparse_rulebook(coords, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    n_out, n_in = nbr.shape[1], len(coords)
    r = np.random.default_rng(2)
    feat = torch.from_numpy(r.normal(size=(n_in, 16)).astype(np.float32)).to(dev).bfloat16()
    w = torch.from_numpy((r.normal(size=(32, 27, 16)) / 20).astype(np.float32)).to(dev).bfloat16()
    nbr_t = torch.from_numpy(np.ascontiguousarray(nbr)).to(dev)
    rcap = L.pvx_tilebook_rcap(16, 32, 27, 2)
    for live in (n_out, 257, 0):
        n_live = torch.tensor([live], dtype=torch.int32, device=dev)
        book = torch.zeros(L.pvx_tilebook_bytes(n_out), dtype=torch.uint8, device=dev)
        _lib.check(L.pvx_tilebook_build(_lib.ptr(nbr_t), 27, n_out, _lib.ptr(n_live), n_in, rcap, _lib.ptr(book), _lib.stream()), "build")
        want, _ = run_conv(dev, feat, w, nbr_t, n_live, 16, 32, 2)
        got, _ = run_conv(dev, feat, w, nbr_t, n_live, 16, 32, 2, book=book, rcap=rcap)
        assert torch.equal(got, want), live
    assert L.pvx_tilebook_rcap(256, 256, 27, 2) == 0 and L.pvx_tilebook_rcap(128, 128, 27, 2) == 320 and L.pvx_tilebook_rcap(16, 16, 27, 0) == 0 and L.pvx_tilebook_rcap(16, 16, 9, 2) == 0
    assert L.pvx_tilebook_build(None, 9, 10, None, 10, 100, None, None) == -1


def test_rolling_form_small_counts_and_long_lines(cuda_device):
    """the 32 -> 32 rolling kernel (its own 896-row window): device-side live counts around the tile size (a workgroup with one
    tile, XCDs without tiles, no next tile to prefetch), and a grid whose y-lines are longer than the window holds, so that
    rows in key order still leave neighbours outside it (the far path beside the register prefetch)"""
    L = _lib.lib()
    dev = cuda_device
    rcap = L.pvx_tilebook_rcap(32, 32, 27, 2)
    assert rcap == 896
    r = np.random.default_rng(4)
    shape = [3, 24, 1500]
    coords = sites(77, 1, shape, 70000, True)                 # ~65 % occupancy: ~1000 sites per y-line
    n = len(coords)
    nbr = loader.subm_rulebook(coords, shape, (3, 3, 3))
    nbr_t = torch.from_numpy(np.ascontiguousarray(nbr)).to(dev)
    feat = torch.from_numpy(r.normal(size=(n, 32)).astype(np.float32)).to(dev).bfloat16()
    w = torch.from_numpy((r.normal(size=(32, 27, 32)) / 30).astype(np.float32)).to(dev).bfloat16()
    res = torch.from_numpy(r.normal(size=(n, 32)).astype(np.float32)).to(dev).bfloat16()
    for live in (n, 0, 1, 255, 256, 257, 600, 256 * 9 + 3):
        n_live = torch.tensor([live], dtype=torch.int32, device=dev)
        book = torch.zeros(L.pvx_tilebook_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(L.pvx_tilebook_build(_lib.ptr(nbr_t), 27, n, _lib.ptr(n_live), n, rcap, _lib.ptr(book), _lib.stream()), "build")
        if live == n:
            _, ent = book_numpy(nbr, n, n, rcap)
            far = (ent == 0xFFFE).mean()
            assert 0.001 < far < 0.5, far                     # key order, and still neighbours beyond 896 rows
        for rr in (None, res):
            want, _ = run_conv(dev, feat, w, nbr_t, n_live, 32, 32, 2, residual=rr)
            got, _ = run_conv(dev, feat, w, nbr_t, n_live, 32, 32, 2, book=book, rcap=rcap, residual=rr)
            assert torch.equal(got, want), (live, rr is None)


@pytest.mark.parametrize("sort", [True, False])
def test_fused_subm_rulebook_and_book(cuda_device, sort):
    """pvx_subm_rulebook_tb = pvx_subm_rulebook + pvx_tilebook_build in one launch: same rulebook, same book bytes (live count
    below the capacity, rows in key order and shuffled, two window sizes)"""
    L = _lib.lib()
    dev = cuda_device
    shape, batch = [10, 50, 300], 2
    coords = sites(13, batch, shape, 40000, sort)
    n = len(coords)
    cap = n + 777
    c4 = np.zeros((cap, 4), np.int32)
    c4[:n] = coords
    coords_t = torch.from_numpy(c4).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    s3 = _lib.int3(shape)
    idx = torch.empty(L.pvx_site_index_bytes(batch, s3), dtype=torch.uint8, device=dev)
    ws = _lib.workspace(L.pvx_site_index_workspace_bytes(batch, s3), dev)
    perm = torch.empty(cap, dtype=torch.int32, device=dev)
    nu = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.pvx_site_index_build(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx), _lib.ptr(perm), _lib.ptr(nu),
                                      _lib.ptr(ws), ws.numel(), _lib.stream()), "index")
    k3, d3 = _lib.int3([3, 3, 3]), _lib.int3([1, 1, 1])
    nbr_a = torch.full((27, cap), -7, dtype=torch.int32, device=dev)
    nbr_b = torch.full((27, cap), -7, dtype=torch.int32, device=dev)
    _lib.check(L.pvx_subm_rulebook(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx), _lib.ptr(perm), k3, d3,
                                   _lib.ptr(nbr_a), _lib.stream()), "rulebook")
    assert np.array_equal(nbr_a[:, :n].cpu().numpy(), loader.subm_rulebook(coords, shape, (3, 3, 3)))
    for rcap in (440, 896):
        book_a = torch.zeros(L.pvx_tilebook_bytes(cap), dtype=torch.uint8, device=dev)
        book_b = torch.zeros(L.pvx_tilebook_bytes(cap), dtype=torch.uint8, device=dev)
        _lib.check(L.pvx_tilebook_build(_lib.ptr(nbr_a), 27, cap, _lib.ptr(n_dev), cap, rcap, _lib.ptr(book_a), _lib.stream()), "build")
        nbr_b.fill_(-7)
        _lib.check(L.pvx_subm_rulebook_tb(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx), _lib.ptr(perm), d3,
                                          _lib.ptr(nbr_b), rcap, _lib.ptr(book_b), _lib.stream()), "fused")
        assert torch.equal(nbr_a, nbr_b)                      # rows past the live count untouched by both
        tiles = (n + 255) // 256
        a = tile_blocks(book_a, tiles)
        b = tile_blocks(book_b, tiles)
        assert np.array_equal(a[:, :7], b[:, :7]) and np.array_equal(a[:, 8:], b[:, 8:])
    assert L.pvx_subm_rulebook_tb(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx), _lib.ptr(perm), d3,
                                  _lib.ptr(nbr_b), 0, _lib.ptr(book_b), None) == -1


def _run_conv_entry(entry, dev, feat, w, nbr_t, n_live_t, cin, cout, code, book, rcap):
    L = _lib.lib()
    cap = nbr_t.shape[1]
    pw = torch.empty(L.pvx_sparse_conv_weight_bytes(cout, 27, cin, code), dtype=torch.uint8, device=dev)
    _lib.check(L.pvx_sparse_conv_prepare_weight(_lib.ptr(w), cout, 27, cin, code, _lib.ptr(pw), _lib.stream()), "prepare")
    out = torch.full((cap, cout), 7.0, dtype=feat.dtype, device=dev)
    scale = torch.linspace(0.5, 1.5, cout, device=dev)
    shift = torch.linspace(-0.2, 0.2, cout, device=dev)
    bias = torch.linspace(-0.1, 0.1, cout, device=dev)
    rc = getattr(L, entry)(_lib.ptr(feat), feat.shape[0], cin, _lib.ptr(pw), _lib.ptr(bias), _lib.ptr(nbr_t), 27, cap, _lib.ptr(n_live_t),
                           cout, code, _lib.ptr(scale), _lib.ptr(shift), None, 1, _lib.ptr(out), _lib.ptr(book), rcap, _lib.stream())
    return rc, out


@pytest.mark.parametrize("sort", [True, False])
def test_clean_index_protocol_and_far_only_book(cuda_device, sort):
    """(1) pvx_site_index_build_clean on a zeroed buffer = pvx_site_index_build; pvx_site_index_unset leaves the buffer all-zero again, also
    when rows were dropped at the capacity (*n_dev > n: the whole index is cleared).  (2) pvx_subm_book stores the same book and exactly
    the far entries of the rulebook, and pvx_sparse_conv_fwd_book over them gives the bits of pvx_sparse_conv_fwd_tb over the full rulebook;
    a shape without a book-form kernel is refused, not served by a kernel that would read the missing entries."""
    L = _lib.lib()
    dev = cuda_device
    shape, batch = [10, 50, 300], 2
    coords = sites(23, batch, shape, 40000, sort)
    n = len(coords)
    cap = n + 500
    c4 = np.zeros((cap, 4), np.int32)
    c4[:n] = coords
    coords_t = torch.from_numpy(c4).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    s3 = _lib.int3(shape)
    nbytes = L.pvx_site_index_bytes(batch, s3)
    ws = _lib.workspace(L.pvx_site_index_workspace_bytes(batch, s3), dev)
    idx_a = torch.full((nbytes,), 0x5a, dtype=torch.uint8, device=dev)
    idx_b = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    perm_a, perm_b = (torch.full((cap,), -3, dtype=torch.int32, device=dev) for _ in range(2))
    nu_a, nu_b = (torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2))
    _lib.check(L.pvx_site_index_build(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx_a), _lib.ptr(perm_a), _lib.ptr(nu_a),
                                      _lib.ptr(ws), ws.numel(), _lib.stream()), "index")
    for rnd in range(2):                                        # the second round runs on the buffer the first round's un-set left
        _lib.check(L.pvx_site_index_build_clean(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx_b), _lib.ptr(perm_b),
                                                _lib.ptr(nu_b), _lib.ptr(ws), ws.numel(), _lib.stream()), "index clean")
        a = idx_a.view(torch.int32).view(-1, 4)
        b = idx_b.view(torch.int32).view(-1, 4)
        used = (a[:, 0] != 0) | (a[:, 1] != 0)
        assert torch.equal(a[:, :2], b[:, :2]) and torch.equal(a[used, 2], b[used, 2])      # bits everywhere, prefix where a bit is set
        assert torch.equal(perm_a, perm_b) and int(nu_a.item()) == int(nu_b.item()) == n
        if rnd == 0:
            _lib.check(L.pvx_site_index_unset(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx_b), _lib.stream()), "unset")
            assert int(idx_b.count_nonzero().item()) == 0
    # rows dropped at the capacity: the caller's coords list is short of the marked cells
    short = n - 1000
    over = torch.tensor([n], dtype=torch.int32, device=dev)
    _lib.check(L.pvx_site_index_unset(_lib.ptr(coords_t), short, _lib.ptr(over), batch, s3, _lib.ptr(idx_b), _lib.stream()), "unset all")
    assert int(idx_b.count_nonzero().item()) == 0
    # strided output index, clean form, and its un-set through coords_out
    k3, st3, p3, d3 = _lib.int3([3, 3, 3]), _lib.int3([2, 2, 2]), _lib.int3([1, 1, 1]), _lib.int3([1, 1, 1])
    o3 = (ctypes.c_int * 3)()
    _lib.check(L.pvx_conv_out_shape(s3, k3, st3, p3, d3, o3), "shape")
    obytes = L.pvx_site_index_bytes(batch, o3)
    ows = _lib.workspace(L.pvx_site_index_workspace_bytes(batch, o3), dev)
    oidx_a = torch.full((obytes,), 0x33, dtype=torch.uint8, device=dev)
    oidx_b = torch.zeros(obytes, dtype=torch.uint8, device=dev)
    no_a, no_b = (torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2))
    for fn, oi, no in ((L.pvx_conv_out_index, oidx_a, no_a), (L.pvx_conv_out_index_clean, oidx_b, no_b)):
        _lib.check(fn(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, k3, st3, p3, d3, _lib.ptr(oi), _lib.ptr(no), _lib.ptr(ows),
                      ows.numel(), _lib.stream()), "out index")
    assert torch.equal(oidx_a.view(torch.int32).view(-1, 4)[:, :3], oidx_b.view(torch.int32).view(-1, 4)[:, :3])
    n_out = int(no_b.item())
    assert n_out == int(no_a.item()) > 0
    ocap = n_out + 10
    ocoords = torch.zeros((ocap, 4), dtype=torch.int32, device=dev)
    onbr = torch.zeros((27, ocap), dtype=torch.int32, device=dev)
    _lib.check(L.pvx_conv_out_rulebook(_lib.ptr(oidx_b), ocap, _lib.ptr(no_b), batch, s3, k3, st3, p3, d3, _lib.ptr(idx_a), _lib.ptr(perm_a),
                                       _lib.ptr(ocoords), _lib.ptr(onbr), _lib.stream()), "out rulebook")
    # the strided rulebook with its tile book in one pass (pvx_conv_out_book), full and far-only
    rc16 = L.pvx_tilebook_rcap(16, 32, 27, _lib.DTYPE_CODE[torch.bfloat16])
    assert rc16 > 0
    for rcap_o in (rc16, 64):                                   # 64 rows: most neighbours outside the window
        obook_a = torch.zeros(L.pvx_tilebook_bytes(ocap), dtype=torch.uint8, device=dev)
        _lib.check(L.pvx_tilebook_build(_lib.ptr(onbr), 27, ocap, _lib.ptr(no_b), cap, rcap_o, _lib.ptr(obook_a), _lib.stream()), "build")
        otiles = (n_out + 255) // 256
        want_book = tile_blocks(obook_a, otiles)
        ocodes = np.ascontiguousarray(want_book[:, 8:]).view(np.uint16).reshape(otiles, 27, 256).transpose(1, 0, 2).reshape(27, otiles * 256)[:, :n_out]
        for far_only in (0, 1):
            obook_b = torch.zeros(L.pvx_tilebook_bytes(ocap), dtype=torch.uint8, device=dev)
            onbr_b = torch.full((27, ocap), -7, dtype=torch.int32, device=dev)
            ocoords_b = torch.zeros((ocap, 4), dtype=torch.int32, device=dev)
            _lib.check(L.pvx_conv_out_book(_lib.ptr(oidx_b), ocap, _lib.ptr(no_b), batch, s3, k3, st3, p3, d3, _lib.ptr(idx_a), _lib.ptr(perm_a),
                                           _lib.ptr(ocoords_b), _lib.ptr(onbr_b), cap, rcap_o, _lib.ptr(obook_b), far_only, _lib.stream()), "out book")
            got_book = tile_blocks(obook_b, otiles)
            assert np.array_equal(want_book[:, :7], got_book[:, :7]) and np.array_equal(want_book[:, 8:], got_book[:, 8:])
            assert torch.equal(ocoords_b, ocoords)
            if far_only:
                farm = ocodes == 0xFFFE
                g_, f_ = onbr_b[:, :n_out].cpu().numpy(), onbr[:, :n_out].cpu().numpy()
                assert np.array_equal(g_[farm], f_[farm]) and np.all(g_[~farm] == -7)
                assert farm.any() or rcap_o == rc16
            else:
                assert torch.equal(onbr_b[:, :n_out], onbr[:, :n_out])
    _lib.check(L.pvx_site_index_unset(_lib.ptr(ocoords), ocap, _lib.ptr(no_b), batch, o3, _lib.ptr(oidx_b), _lib.stream()), "unset out")
    assert int(oidx_b.count_nonzero().item()) == 0

    # ---- far-only rulebook ----------------------------------------------------------------------------------------
    cin, cout, code = 64, 64, _lib.DTYPE_CODE[torch.bfloat16]
    rcap = L.pvx_tilebook_rcap(cin, cout, 27, code)
    assert rcap > 0
    nbr_full = torch.full((27, cap), -7, dtype=torch.int32, device=dev)
    nbr_far = torch.full((27, cap), -7, dtype=torch.int32, device=dev)
    book_a = torch.zeros(L.pvx_tilebook_bytes(cap), dtype=torch.uint8, device=dev)
    book_b = torch.zeros(L.pvx_tilebook_bytes(cap), dtype=torch.uint8, device=dev)
    _lib.check(L.pvx_subm_rulebook_tb(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx_a), _lib.ptr(perm_a), d3,
                                      _lib.ptr(nbr_full), rcap, _lib.ptr(book_a), _lib.stream()), "fused")
    _lib.check(L.pvx_subm_book(_lib.ptr(coords_t), cap, _lib.ptr(n_dev), batch, s3, _lib.ptr(idx_a), _lib.ptr(perm_a), d3,
                               _lib.ptr(nbr_far), rcap, _lib.ptr(book_b), _lib.stream()), "book")
    tiles = (n + 255) // 256
    a = tile_blocks(book_a, tiles)
    b = tile_blocks(book_b, tiles)
    assert np.array_equal(a[:, :7], b[:, :7]) and np.array_equal(a[:, 8:], b[:, 8:])
    codes = np.ascontiguousarray(a[:, 8:]).view(np.uint16).reshape(tiles, 27, 256).transpose(1, 0, 2).reshape(27, tiles * 256)[:, :n]
    far = codes == 0xFFFE
    got_far, full = nbr_far[:, :n].cpu().numpy(), nbr_full[:, :n].cpu().numpy()
    assert np.array_equal(got_far[far], full[far]) and np.all(got_far[~far] == -7)     # exactly the far entries, nothing else
    assert far.any() == (not sort)                              # key order at this density: every neighbour inside the window
    g = torch.Generator().manual_seed(5)
    feat = torch.randn((cap, cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((cout, 27, cin), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    rc_a, out_a = _run_conv_entry("pvx_sparse_conv_fwd_tb", dev, feat, w, nbr_full, n_dev, cin, cout, code, book_a, rcap)
    rc_b, out_b = _run_conv_entry("pvx_sparse_conv_fwd_book", dev, feat, w, nbr_far, n_dev, cin, cout, code, book_b, rcap)
    assert rc_a == 0 and rc_b == 0
    assert torch.equal(out_a, out_b) and float(out_a[:n].float().abs().sum()) > 0
    # no book-form kernel for this request: refused (PVX_ERR_RANGE / PVX_ERR_INVALID), not served from the incomplete rulebook
    w128 = torch.zeros((128, 27, 128), dtype=torch.bfloat16, device=dev)
    f128 = torch.zeros((cap, 128), dtype=torch.bfloat16, device=dev)
    rc, _ = _run_conv_entry("pvx_sparse_conv_fwd_book", dev, f128, w128, nbr_far, n_dev, 128, 128, code, book_b, rcap)
    assert rc == -4
    rc, _ = _run_conv_entry("pvx_sparse_conv_fwd_book", dev, feat, w, nbr_far, n_dev, cin, cout, code, None, 0)
    assert rc == -1
    # the host query a plan uses before it stores a rulebook far-only
    assert L.pvx_sparse_conv_book_form(cap, 64, 64, 27, cap, code, rcap) == 1
    assert L.pvx_sparse_conv_book_form(cap, 64, 64, 27, cap, code, rcap + 1) == 0
    assert L.pvx_sparse_conv_book_form(cap, 128, 128, 27, cap, code, rcap) == 0
    assert L.pvx_sparse_conv_book_form(1 << 24, 64, 64, 27, cap, code, rcap) == 0        # 2 GiB of input rows: 32-bit buffer offsets
    assert L.pvx_sparse_conv_book_form(cap, 16, 32, 27, cap, code, L.pvx_tilebook_rcap(16, 32, 27, code)) == 1
    assert L.pvx_sparse_conv_book_form(cap, 64, 64, 27, cap, 0, rcap) == 0

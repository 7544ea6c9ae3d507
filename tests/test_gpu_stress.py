"""Randomised shape sweep of the kernels behind the C ABI against direct NumPy
statements (seeded; a few dozen small problems per kernel).  Complements the
golden-vector tests: odd voxel counts, one-row cells, k not a multiple of 16,
item counts that do not fill a workgroup's item group, n on both sides of the
4-wave / 8-wave switch of the latent kernel."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cells(rs, nz, ncell):
    cuts = np.sort(rs.choice(np.arange(1, nz), size=ncell - 1, replace=False)) if ncell > 1 else np.array([], int)
    return np.concatenate(([0], cuts, [nz])).astype(np.int64)


def _zscore_items(X, src, cell_lo, zflags):
    Z = np.empty((src.shape[0], src.shape[1], X.shape[1]))
    for b in range(src.shape[0]):
        G = X[src[b]]
        for c, (lo, hi) in enumerate(zip(cell_lo[:-1], cell_lo[1:])):
            if zflags[c]:
                blk = G[lo:hi]
                mu, sd = blk.mean(0), blk.std(0)
                with np.errstate(divide="ignore", invalid="ignore"):
                    z = (blk - mu) / sd / np.sqrt(hi - lo)
                z[:, sd <= 2.220446049250313e-16 * np.abs(mu)] = 0.0
                Z[b, lo:hi] = z
            else:
                Z[b, lo:hi] = G[lo:hi]
    return Z


def test_fused_items_random_shapes():
    import torch
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(11)
    for trial in range(24):
        n = int(rs.randint(5, 140))
        p = int(rs.choice([1, 15, 64, 65, 129, 300]))
        nz = int(rs.randint(4, 150))
        ncell = int(rs.randint(1, min(nz, 9)))
        k = int(rs.choice([1, 5, 16, 17, 33, 48, 70]))
        items = int(rs.randint(1, 8))
        cell_lo = _cells(rs, nz, ncell)
        zflags = rs.randint(0, 2, size=ncell)
        X = rs.randn(n, p) * 2 + rs.randn(1, p) * 10
        src = rs.randint(0, n, size=(items, nz)).astype(np.int32)
        rows = rs.randn(items, k, nz)
        ref = rs.randn(p, k)
        eng = ProjectionEngine(X)
        S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
        S2 = torch.zeros_like(S1)
        vst, rowsq = eng.item_fused(src, cell_lo, zflags, rows, ref=ref, S1=S1, S2=S2, want_vst=True,
                                    want_rowsq=True)
        Z = _zscore_items(X, src, cell_lo, zflags)
        want = np.einsum("bji,biv->bjv", rows, Z)
        scale = max(np.abs(want).max(), 1e-300)
        tag = f"trial {trial}: n={n} p={p} nz={nz} cells={ncell} k={k} items={items}"
        np.testing.assert_allclose(vst.cpu().numpy(), want, rtol=1e-9, atol=1e-10 * scale, err_msg=tag)
        np.testing.assert_allclose(rowsq.cpu().numpy(), (want ** 2).sum(-1), rtol=1e-9, atol=1e-18 * scale ** 2,
                                   err_msg=tag)
        d = np.transpose(want, (0, 2, 1)) - ref
        np.testing.assert_allclose(S1.cpu().numpy(), d.sum(0), rtol=1e-9, atol=1e-9 * (scale + 1), err_msg=tag)
        np.testing.assert_allclose(S2.cpu().numpy(), (d ** 2).sum(0), rtol=1e-9, atol=1e-9 * (scale + 1) ** 2,
                                   err_msg=tag)


def _check_items(eng, X, src, cell_lo, zflags, rows, ref, tag, ranges, expect):
    import torch
    p, k = X.shape[1], rows.shape[1]
    S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
    S2 = torch.zeros_like(S1)
    vst, rowsq = eng.item_fused(src, cell_lo, zflags, rows, ref=ref, S1=S1, S2=S2, want_vst=True, want_rowsq=True,
                                src_ranges=ranges)
    assert eng.last_item_kernel == expect, tag
    Z = _zscore_items(X, src, cell_lo, zflags)
    want = np.einsum("bji,biv->bjv", rows, Z)
    scale = max(np.abs(want).max(), 1e-300)
    np.testing.assert_allclose(vst.cpu().numpy(), want, rtol=1e-9, atol=1e-10 * scale, err_msg=tag)
    np.testing.assert_allclose(rowsq.cpu().numpy(), (want ** 2).sum(-1), rtol=1e-9, atol=1e-18 * scale ** 2,
                               err_msg=tag)
    d = np.transpose(want, (0, 2, 1)) - ref
    np.testing.assert_allclose(S1.cpu().numpy(), d.sum(0), rtol=1e-9, atol=1e-9 * (scale + 1), err_msg=tag)
    np.testing.assert_allclose(S2.cpu().numpy(), (d ** 2).sum(0), rtol=1e-9, atol=1e-9 * (scale + 1) ** 2,
                               err_msg=tag)


def test_aggregated_items_random_shapes():
    """K4a (plsr_item_agg: aggregated operators, X in registers, statistics by MFMA) against
    NumPy on items whose cells draw from their own row ranges, as the reference's bootstrap
    does: aligned and unaligned cells (several sweeps), cells of one and two rows whose samples
    are constant (duplicates), a copied block over all rows next to z-scored cells (the
    multiblock's shape), item counts off the group of four, k in all three tile counts."""
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(77)
    for trial in range(28):
        n = int(rs.choice([6, 17, 22, 40, 60, 64, 96, 120, 128]))
        p = int(rs.choice([1, 15, 64, 65, 129, 300]))
        k = int(rs.choice([1, 5, 16, 17, 33, 48]))
        items = int(rs.randint(1, 14))
        aligned = trial % 3 != 0
        # z-scored cells: a partition of (a subset of) the source rows
        ncz = int(rs.randint(1, min(n // 4 if aligned else n, 7) + 1))
        if aligned:
            cuts = 4 * np.sort(rs.choice(np.arange(1, n // 4 + 1), size=ncz, replace=False))
            bounds = np.concatenate(([0], cuts))
        else:
            bounds = _cells(rs, n, ncz)
        task = trial % 4 == 1                  # multiblock: copied block over all rows first
        src_lo, src_hi, cell_lo, zflags, cols = [], [], [0], [], []
        if task:
            src_lo.append(0), src_hi.append(n), zflags.append(0)
            cell_lo.append(n)
            cols.append(rs.randint(0, n, size=(items, n)))
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            if rs.rand() < 0.2 and len(bounds) > 2:
                continue                        # rows no cell reads
            cnt = int(hi - lo) if rs.rand() < 0.7 else int(rs.randint(1, 2 * (hi - lo) + 1))
            src_lo.append(int(lo)), src_hi.append(int(hi)), zflags.append(1)
            cell_lo.append(cell_lo[-1] + cnt)
            cols.append(rs.randint(lo, hi, size=(items, cnt)))
        if not cols:
            continue
        src = np.concatenate(cols, axis=1).astype(np.int32)
        cell_lo = np.array(cell_lo)
        nz = int(cell_lo[-1])
        X = rs.randn(n, p) * 2 + rs.randn(1, p) * 10 + 5.0 * (np.arange(n)[:, None] // 7)
        rows = rs.randn(items, k, nz)
        ref = rs.randn(p, k)
        eng = ProjectionEngine(X)
        tag = f"trial {trial}: n={n} p={p} nz={nz} cells={len(zflags)} k={k} items={items} aligned={aligned}"
        ranges = (np.array(src_lo), np.array(src_hi))
        found = eng.source_ranges(src, cell_lo)
        assert (found[0] >= ranges[0]).all() and (found[1] <= ranges[1]).all(), tag
        need = eng.lib.plsr_item_agg_workspace_bytes(
            n, nz, k, (ctypes.c_int32 * len(cell_lo))(*cell_lo), (ctypes.c_int32 * len(zflags))(*zflags),
            (ctypes.c_int32 * len(zflags))(*src_lo), (ctypes.c_int32 * len(zflags))(*src_hi), len(zflags), items, p,
            1, 1)
        _check_items(eng, X, src, cell_lo, np.array(zflags), rows, ref, tag, ranges, "agg" if need else "gather")


def test_two_stage_behaviour_items_random_shapes():
    """K4b (plsr_item_beh: behaviour PLS in two stages, two items per stage-1 MFMA tile, scaled
    accumulators as the B operand of the projection) against NumPy: cells of unequal, unaligned
    sizes in all four register layouts, 1 .. 16 behaviours, samples larger and smaller than their
    source range, constant samples in small cells, item counts off the pairs and the groups of four."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(31)
    shapes = [(6, 20, 8), (6, 17, 3), (4, 31, 8), (3, 9, 16), (8, 13, 5), (8, 16, 1), (12, 7, 4), (16, 5, 2),
              (2, 2, 2), (1, 30, 11), (5, 8, 9)]
    for trial, (ncell, crow, b) in enumerate(shapes):
        sizes = rs.randint(max(1, crow - 3), crow + 1, size=ncell)          # source rows per cell
        src_lo = np.concatenate(([0], np.cumsum(sizes)[:-1])) + rs.randint(0, 3)
        src_hi = src_lo + sizes
        n = int(src_hi[-1] + rs.randint(0, 4))
        p = int(rs.choice([1, 15, 64, 65, 130, 257]))
        items = int(rs.randint(1, 12))
        k = min(48, ncell * b)
        cnts = [int(sz if rs.rand() < 0.6 else rs.randint(1, 2 * sz + 1)) for sz in sizes]
        cell_lo = np.concatenate(([0], np.cumsum(cnts)))
        nz = int(cell_lo[-1])
        src = np.concatenate([rs.randint(lo, hi, size=(items, c)) for lo, hi, c in zip(src_lo, src_hi, cnts)],
                             axis=1).astype(np.int32)
        X = rs.randn(n, p) * 2 + rs.randn(1, p) * 10 + 3.0 * (np.arange(n)[:, None] // 5)
        Y = rs.randn(items, nz, b)
        Yz = np.empty_like(Y)
        for lo, hi in zip(cell_lo[:-1], cell_lo[1:]):                       # z-scored within cells (sum 0)
            blk = Y[:, lo:hi]
            sd = blk.std(1, keepdims=True)
            Yz[:, lo:hi] = np.where(sd > 0, (blk - blk.mean(1, keepdims=True)) / np.where(sd > 0, sd, 1), 0.0)
        U = rs.randn(ncell * b, k)
        ref = rs.randn(p, k)
        eng = ProjectionEngine(X)
        tag = f"trial {trial}: cells={ncell} rows~{crow} b={b} n={n} p={p} nz={nz} k={k} items={items}"
        S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
        S2 = torch.zeros_like(S1)
        vst = eng.item_beh(src, cell_lo, (src_lo, src_hi), Yz, U, ref=ref, S1=S1, S2=S2)
        assert vst is not None, tag
        rows = np.zeros((items, k, nz))
        for c, (lo, hi) in enumerate(zip(cell_lo[:-1], cell_lo[1:])):
            rows[:, :, lo:hi] = np.einsum("bie,ej->bji", Yz[:, lo:hi], U[c * b:(c + 1) * b])
        Z = _zscore_items(X, src, cell_lo, np.ones(ncell, int))
        want = np.einsum("bji,biv->bjv", rows, Z)
        scale = max(np.abs(want).max(), 1e-300)
        np.testing.assert_allclose(vst.cpu().numpy(), want, rtol=1e-9, atol=1e-10 * scale, err_msg=tag)
        d = np.transpose(want, (0, 2, 1)) - ref
        np.testing.assert_allclose(S1.cpu().numpy(), d.sum(0), rtol=1e-9, atol=1e-9 * (scale + 1), err_msg=tag)
        np.testing.assert_allclose(S2.cpu().numpy(), (d ** 2).sum(0), rtol=1e-9, atol=1e-9 * (scale + 1) ** 2,
                                   err_msg=tag)
        # the same VS^T tile-major (K5i's operand layout): [item][tile of 32 voxels][k][32], bit for bit
        tl = eng.item_beh(src, cell_lo, (src_lo, src_hi), Yz, U, tiled=True).cpu().numpy()
        ppad = (p + 31) // 32 * 32
        assert tl.shape == (items, k, ppad), tag
        back = tl.reshape(items, ppad // 32, k, 32).transpose(0, 2, 1, 3).reshape(items, k, ppad)[:, :, :p]
        np.testing.assert_array_equal(back, vst.cpu().numpy(), err_msg=tag)
    # a row outside its cell's range poisons its item
    src[0, 0] = (src_hi[0] + 1) % n if ncell > 1 else src[0, 0]
    if ncell > 1 and not (src_lo[0] <= src[0, 0] < src_hi[0]):
        got = eng.item_beh(src, cell_lo, (src_lo, src_hi), Yz, U).cpu().numpy()
        assert np.isnan(got[0]).any()


def test_aggregated_items_reject_rows_outside_their_range():
    """An item that reads a source row outside its cell's declared range is returned as NaN."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(5)
    n, p, k = 24, 70, 5
    X = rs.randn(n, p)
    cell_lo = np.array([0, 12, 24])
    src = np.stack([np.concatenate((rs.randint(0, 12, 12), rs.randint(12, 24, 12))) for _ in range(6)]).astype(np.int32)
    src[4, 3] = 17
    eng = ProjectionEngine(X)
    S1 = torch.zeros((p, k), dtype=torch.float64, device=eng.device)
    S2 = torch.zeros_like(S1)
    vst, _ = eng.item_fused(src, cell_lo, np.ones(2, int), rs.randn(6, k, 24), S1=S1, S2=S2, want_vst=True,
                            src_ranges=(np.array([0, 12]), np.array([12, 24])))
    assert eng.last_item_kernel == "agg"
    got = vst.cpu().numpy()
    assert np.isnan(got[4]).any() and not np.isnan(got[[0, 1, 2, 3, 5]]).any()
    assert np.isnan(S1.cpu().numpy()).any()


def test_fused_gram_and_latent_random_shapes():
    import torch
    from plspy_amd import _lib
    from plspy_amd.engine import ProjectionEngine, _ptr, _stream
    rs = np.random.RandomState(12)
    for trial in range(16):
        n = int(rs.choice([6, 17, 48, 64, 65, 100, 130]))
        p = int(rs.choice([7, 64, 200, 1000]))
        nz = int(rs.randint(4, 120))
        ncell = int(rs.randint(1, min(nz, 7)))
        m = int(rs.choice([1, 9, 16, 30, 50]))
        items = int(rs.randint(1, 8))
        cell_lo = _cells(rs, nz, ncell)
        zflags = rs.randint(0, 2, size=ncell)
        X = rs.randn(n, p) + 3.0
        src = rs.randint(0, n, size=(items, nz)).astype(np.int32)
        rows = rs.randn(items, m, nz)
        eng = ProjectionEngine(X)
        tag = f"trial {trial}: n={n} p={p} nz={nz} cells={ncell} m={m} items={items}"
        # Gram with the gather / z-score fused into its staging
        G = eng.gram_phase(rows, gather=dict(src=src, cell_lo=cell_lo, cell_z=zflags)).cpu().numpy()
        M = np.einsum("bji,biv->bjv", rows, _zscore_items(X, src, cell_lo, zflags))
        want = np.einsum("bjv,blv->bjl", M, M)
        np.testing.assert_allclose(G[:, :m, :m], want, rtol=1e-9, atol=1e-10 * np.abs(want).max(), err_msg=tag)
        # latent kernel: Zt = VS X^T, nsq = row norms^2 of VS
        k = min(m, 64)
        vs = rs.randn(items, k, p)
        d_vs = eng.dev(vs)
        need = eng.lib.plsr_latent_workspace_bytes(n, k, items, p)
        assert need > 0, tag
        work = torch.empty(need, dtype=torch.uint8, device=eng.device)
        Zt = torch.empty((items, k, n), dtype=torch.float64, device=eng.device)
        nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device)
        _lib.check(eng.lib.plsr_latent(_ptr(eng.X), eng.X.stride(0), p, n, _ptr(d_vs), p, items, k, _ptr(Zt),
                                       _ptr(nsq), _ptr(work), need, _stream()), "plsr_latent")
        wantz = np.einsum("bjv,iv->bji", vs, X)
        np.testing.assert_allclose(Zt.cpu().numpy(), wantz, rtol=1e-9, atol=1e-10 * np.abs(wantz).max(), err_msg=tag)
        np.testing.assert_allclose(nsq.cpu().numpy(), (vs ** 2).sum(-1), rtol=1e-10, err_msg=tag)


def test_latent_batch_random_shapes():
    """engine.latent_batch against NumPy: K5x (X read as pre-transposed B fragments, n <= 128) and the
    LDS-staged K5 beyond; voxel counts off the 32-voxel tile, k off the 16-row tile, 4- and 8-wave shapes."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(41)
    for n, p, k, items in [(5, 1, 1, 1), (60, 257, 6, 3), (64, 1000, 17, 2), (65, 333, 48, 5), (120, 2001, 38, 4),
                           (128, 96, 64, 2), (17, 31, 33, 7), (130, 515, 12, 3), (200, 64, 48, 2)]:
        X = rs.randn(n, p) + 2.0
        vs = rs.randn(items, k, p)
        eng = ProjectionEngine(X)
        Zt = torch.empty((items, k, n), dtype=torch.float64, device=eng.device)
        nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device)
        eng.latent_batch(eng.dev(vs), n, Zt, nsq)
        want = np.einsum("bjv,iv->bji", vs, X)
        tag = f"n={n} p={p} k={k} items={items}"
        np.testing.assert_allclose(Zt.cpu().numpy(), want, rtol=1e-11, atol=1e-11 * np.abs(want).max(), err_msg=tag)
        np.testing.assert_allclose(nsq.cpu().numpy(), (vs ** 2).sum(-1), rtol=1e-11, err_msg=tag)
        eng.latent_batch(eng.dev(vs), n, Zt)                       # without the norms
        np.testing.assert_allclose(Zt.cpu().numpy(), want, rtol=1e-11, atol=1e-11 * np.abs(want).max(), err_msg=tag)


def test_projection_phases_random_shapes():
    """K1 through the engine: permutation norms, bootstrap moments / norms / T and
    the dumped VS for random n, k (all periods incl. padded ones), R, p."""
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(13)
    # the last trials have larger n: from n > 72 the LDS-fed permutation kernel, from n > 144 the
    # bootstrap kernel too, run eight waves per workgroup, up to the largest n whose tile fits (256)
    big_n = [73, 100, 120, 144, 145, 150, 197, 240, 240, 252, 256, 256]
    for trial in range(32):
        n = int(rs.randint(3, 70)) if trial < 20 else big_n[trial - 20]
        p = int(rs.choice([1, 63, 64, 65, 257]))
        k = int(rs.randint(1, 15))
        k2 = int(rs.randint(0, 13))
        R = int(rs.randint(1, 30)) if trial < 20 else int(rs.randint(20, 90))
        X = rs.randn(n, p)
        M = rs.randn(n, k)
        inds = rs.randint(0, n, size=(R, n)).astype(np.int32)
        Xm = rs.randn(k2, p) if k2 else None
        ref = rs.randn(p, k)
        eng = ProjectionEngine(X)
        tag = f"trial {trial}: n={n} p={p} k={k} k2={k2} R={R}"
        # VS_b = X^T Op_b with Op_b[i] = sum_{r: inds[b,r]=i} M[r]
        Op = np.zeros((R, n, k))
        for b in range(R):
            np.add.at(Op[b], inds[b], M)
        VS = np.einsum("iv,bik->bvk", X, Op)
        ssq = eng.perm_phase(k, inds=inds, M=M).cpu().numpy()
        np.testing.assert_allclose(ssq, (VS ** 2).sum(1), rtol=1e-10, atol=1e-12, err_msg=tag)
        res = eng.boot_phase(k, inds=inds, M=M, ref=ref, Xm=eng.dev(Xm) if k2 else None, dump=True)
        np.testing.assert_allclose(res["vs"].cpu().numpy(), VS, rtol=1e-10, atol=1e-12, err_msg=tag)
        np.testing.assert_allclose(res["ssq"].cpu().numpy(), (VS ** 2).sum(1), rtol=1e-10, atol=1e-12, err_msg=tag)
        d = VS - ref
        np.testing.assert_allclose(res["S1"].cpu().numpy(), d.sum(0), rtol=1e-9, atol=1e-10, err_msg=tag)
        np.testing.assert_allclose(res["S2"].cpu().numpy(), (d ** 2).sum(0), rtol=1e-9, atol=1e-10, err_msg=tag)
        if k2:
            T = np.einsum("bvk,cv->bkc", VS, Xm)
            np.testing.assert_allclose(res["T"].cpu().numpy(), T, rtol=1e-9, atol=1e-10 * (np.abs(T).max() + 1),
                                       err_msg=tag)


def test_apply_rows_random_shapes():
    """K0 (plsr_apply_rows: the observed blocks, rows @ X) against NumPy: row counts on both
    sides of the 16-row slice, voxel counts that do not fill a workgroup, n not a multiple of 4."""
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(99)
    for n, p, m in [(7, 1, 1), (60, 255, 6), (61, 257, 16), (120, 1000, 17), (240, 513, 40), (13, 64, 33)]:
        X = rs.randn(n, p)
        rows = rs.randn(m, n)
        eng = ProjectionEngine(X)
        got = eng.apply_operator(rows).cpu().numpy()
        want = rows @ X
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max(), err_msg=f"{n} {p} {m}")
    with pytest.raises(ValueError):
        eng.apply_operator(np.zeros((2, 5)))


def test_latents_random_shapes():
    """engine.latents (observed X @ V through K5 with one item) against NumPy: n on both sides of
    the 4-wave / 8-wave switch, k up to the kernel's 64, voxel counts off the 32-voxel tile."""
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(21)
    for n, p, k in [(5, 1, 1), (60, 257, 6), (64, 1000, 17), (65, 333, 48), (120, 2001, 38), (200, 515, 64),
                    (256, 96, 12)]:
        X = rs.randn(n, p)
        V = rs.randn(p, k)
        got = ProjectionEngine(X).latents(V)
        want = X @ V
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11 * np.abs(want).max(), err_msg=f"{n} {p} {k}")


@pytest.mark.gpu
def test_scale_cols_matches_numpy():
    """plsr_scale_cols: V * s for host and device inputs (the device input is left untouched)."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(3)
    eng = ProjectionEngine(rs.randn(4, 70))
    for rows, cols in ((1, 1), (1001, 7), (5000, 48)):
        V, s = rs.randn(rows, cols), rs.randn(cols)
        assert np.array_equal(eng.scale_cols(V, s).cpu().numpy(), V * s)
        Vd = torch.as_tensor(V).cuda()
        assert np.array_equal(eng.scale_cols(Vd, torch.as_tensor(s).cuda()).cpu().numpy(), V * s)
        assert np.array_equal(Vd.cpu().numpy(), V)


@pytest.mark.gpu
def test_fused_gram_block_sparse_operators():
    """plsr_gram_fused with operators that are zero in whole (tile, k-step) blocks: the kernel skips the
    k-steps of a tile group that gram_activity_kernel finds empty.  Shapes of every tile-group split
    (m = 17 .. 96: one to six tiles), split-half-like halves, halves that overlap, an item that is dense
    while the others are sparse (the masks are a union over the launch's items), an empty group."""
    from plspy_amd.engine import ProjectionEngine
    rs = np.random.RandomState(21)
    for trial, m in enumerate([17, 32, 33, 40, 48, 50, 64, 65, 76, 80, 90, 96, 76, 96, 24]):
        n, p = 40, int(rs.choice([130, 777, 2000]))
        nz = int(rs.choice([40, 75, 100, 200]))
        ncell = int(rs.randint(1, 7))
        items = int(rs.randint(1, 6))
        cell_lo = _cells(rs, nz, ncell)
        zflags = rs.randint(0, 2, size=ncell)
        X = rs.randn(n, p) + 1.0
        src = rs.randint(0, n, size=(items, nz)).astype(np.int32)
        rows = rs.randn(items, m, nz)
        half, cut = m // 2, int(rs.randint(1, nz))
        mode = trial % 5
        if mode == 0:                               # two halves on disjoint rows (split-half)
            rows[:, :half, cut:] = 0.0
            rows[:, half:, :cut] = 0.0
        elif mode == 1:                             # halves whose row ranges overlap
            rows[:, :half, min(nz, cut + 9):] = 0.0
            rows[:, half:, :max(0, cut - 9)] = 0.0
        elif mode == 2:                             # the last item dense, the others sparse
            rows[:-1, :half, cut:] = 0.0
            rows[:-1, half:, :cut] = 0.0
        elif mode == 3:                             # the first half does nothing at all
            rows[:, :half] = 0.0
        else:                                       # scattered zero blocks of four rows
            for s0 in range(0, nz, 4):
                if rs.rand() < 0.5:
                    rows[:, :half, s0:s0 + 4] = 0.0
                if rs.rand() < 0.5:
                    rows[:, half:, s0:s0 + 4] = 0.0
        eng = ProjectionEngine(X)
        tag = f"trial {trial}: m={m} nz={nz} cells={ncell} items={items} p={p} mode={mode}"
        G = eng.gram_phase(rows, gather=dict(src=src, cell_lo=cell_lo, cell_z=zflags)).cpu().numpy()
        M = np.einsum("bji,biv->bjv", rows, _zscore_items(X, src, cell_lo, zflags))
        want = np.einsum("bjv,blv->bjl", M, M)
        np.testing.assert_allclose(G[:, :m, :m], want, rtol=1e-9, atol=1e-10 * max(np.abs(want).max(), 1e-300),
                                   err_msg=tag)


def test_latent_of_an_x_beyond_4_gib():
    """K5 addresses the rows it stages by 32-bit byte offsets; an X of 4 GiB or more (here 240 x 2 300 000
    doubles = 4.4 GB, made on the device) is walked in blocks of rows (plsr_latent).  Against torch's own
    fp64 product."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    n, p, k, items = 240, 2_300_000, 12, 2
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=g)
    vs = torch.randn((items, k, p), dtype=torch.float64, device="cuda", generator=g)
    eng = ProjectionEngine(X)
    Zt = torch.empty((items, k, n), dtype=torch.float64, device=eng.device)
    nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device)
    eng.latent_batch(vs, n, Zt, nsq)
    want = torch.matmul(vs, X.t())
    scale = float(want.abs().max())
    assert float((Zt - want).abs().max()) <= 1e-11 * scale
    assert torch.allclose(nsq, (vs * vs).sum(-1), rtol=1e-11)
    del X, vs, want
    torch.cuda.empty_cache()


@pytest.mark.parametrize("tiled", [False, True])
def test_latent_by_index_of_a_vs_beyond_2_gib(tiled):
    """K5i reaches VS^T through buffer descriptors with 32-bit offsets: an item of 64 x 5 000 000 doubles (2.56 GB:
    past the 2 GiB a signed record count would cover, below the 4 GiB the library accepts), row-major and
    tile-major, against torch's own fp64 product."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    n, p, k, items, m = 24, 5_000_000, 64, 1, 24
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=g)
    vs = torch.randn((items, k, p), dtype=torch.float64, device="cuda", generator=g)
    eng = ProjectionEngine(X)
    idx = np.random.RandomState(2).randint(0, n, size=(items, m)).astype(np.int32)
    d_idx = eng.dev(idx, torch.int32)
    want = torch.matmul(vs, X[torch.as_tensor(idx[0], device="cuda").long()].t())
    d_vs = vs.view(items, k, p // 32, 32).transpose(1, 2).contiguous().view(items, k, p) if tiled else vs
    L = torch.empty((items, k, m), dtype=torch.float64, device=eng.device)
    nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device)
    eng.latent_batch_index(d_vs, n, idx, d_idx, L, nsq, tiled=tiled)
    assert eng.last_latent_kernel == "index"
    assert float((L - want).abs().max()) <= 1e-11 * float(want.abs().max())
    assert torch.allclose(nsq, (vs * vs).sum(-1), rtol=1e-11)
    del X, vs, want, d_vs
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [
    # kr = k, p, items
    (38, 2001, 9), (5, 130, 3), (16, 64, 17), (33, 777, 4), (48, 515, 6), (1, 31, 2),
])
def test_rows_project_matches_numpy(shape):
    """K4m (plsr_rows_project): VS_b = (U^T D_b^-1) R_b in place + shifted moment sums, against NumPy; a row of
    norm 0, voxel counts off the tile, k off the 16-row tile."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    kr, p, items = shape
    rs = np.random.RandomState(kr + p)
    eng = ProjectionEngine(rs.randn(4, p))
    R = rs.randn(items, kr, p) * np.exp(rs.randn(items, kr, 1))
    rowsq = (R ** 2).sum(-1) * (0.5 + rs.rand(items, kr))      # (any positive numbers: the kernel takes them as given)
    if kr > 2:
        rowsq[0, 1] = 0.0                                        # a row of norm 0 contributes 0
    U = np.linalg.qr(rs.randn(kr, kr))[0]
    ref = rs.randn(p, kr)
    d_R = eng.dev(R.copy())
    k16 = (kr + 15) // 16 * 16
    d_sq = torch.zeros((items, k16), dtype=torch.float64, device=eng.device)
    d_sq[:, :kr] = eng.dev(rowsq)
    S1 = torch.zeros((p, kr), dtype=torch.float64, device=eng.device)
    S2 = torch.zeros_like(S1)
    # to a block of its own first (R stays), then in place: the same bits
    d_out = torch.full((items, kr, p), float("nan"), dtype=torch.float64, device=eng.device)
    assert eng.rows_project(d_R, d_sq, U, out=d_out)
    np.testing.assert_array_equal(d_R.cpu().numpy(), R)
    assert eng.rows_project(d_R, d_sq, U, ref=ref, S1=S1, S2=S2)
    np.testing.assert_array_equal(d_out.cpu().numpy(), d_R.cpu().numpy())
    with np.errstate(divide="ignore"):
        inv = np.where(rowsq > 0, 1.0 / np.sqrt(rowsq), 0.0)
    want = np.einsum("rj,br,brv->bjv", U, inv, R)
    scale = np.abs(want).max()
    np.testing.assert_allclose(d_R.cpu().numpy(), want, rtol=1e-11, atol=1e-12 * scale)
    d = np.transpose(want, (0, 2, 1)) - ref
    np.testing.assert_allclose(S1.cpu().numpy(), d.sum(0), rtol=1e-10, atol=1e-11 * scale * items)
    np.testing.assert_allclose(S2.cpu().numpy(), (d ** 2).sum(0), rtol=1e-10, atol=1e-11 * scale ** 2 * items)


@pytest.mark.parametrize("shape", [
    # n, k, p, items, m, kind of index rows
    (120, 48, 3001, 9, 120, "boot"), (120, 32, 2050, 5, 200, "boot2"), (128, 16, 515, 4, 128, "perm"),
    (60, 12, 777, 6, 60, "boot"), (33, 5, 100, 3, 50, "boot"), (97, 64, 1030, 3, 97, "few"),
    (120, 38, 2050, 5, 200, "boot2"), (120, 48, 200_003, 3, 120, "boot"), (128, 64, 700, 3, 128, "perm"),
    (140, 16, 300, 2, 140, "boot-full"), (16, 4, 20, 1, 16, "boot"), (5, 3, 33, 2, 9, "boot"),
])
def test_latent_by_index_matches_numpy(shape):
    """K5i (plsr_latent_index): L_b = (X[idx_b] VS_b^T)^T computed on the different rows of each sample
    only, against NumPy: bootstrap samples (about 0.63 n different rows, the count differs from item to item),
    two samples side by side (m > n), a permutation (every row, n = 128: eight row tiles), a sample of three rows, a
    long X (voxel ranges of many tiles), more row tiles x tiles of latent variables than a wave holds (k = 38 with
    seven row tiles, k = 64 with eight: the second launch); with and without the column norms.  "boot-full": more than
    128 rows -- the engine then takes the full product and gathers its columns."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    n, k, p, items, m, kind = shape
    rs = np.random.RandomState(n + k + p)
    X = rs.randn(n, p)
    eng = ProjectionEngine(X)
    vs = rs.randn(items, k, p)
    if kind == "perm":
        idx = np.stack([rs.permutation(n)[:m] for _ in range(items)])
    elif kind == "few":
        idx = rs.choice(rs.choice(n, 3, replace=False), size=(items, m))
    else:
        idx = rs.randint(0, n, size=(items, m))
    idx = idx.astype(np.int32)
    d_vs, d_idx = eng.dev(vs), eng.dev(idx, torch.int32)
    want = np.einsum("bjv,biv->bji", vs, X[idx])
    ppad = (p + 31) // 32 * 32
    vt = np.full((items, k, ppad), np.nan)                     # (the padding of the last tile is never read as data)
    vt[:, :, :p] = vs
    d_vt = eng.dev(np.ascontiguousarray(vt.reshape(items, k, ppad // 32, 32).transpose(0, 2, 1, 3)).reshape(items, k, ppad))
    for with_norms in (False, True):
        for tiled in ((False,) if kind.endswith("full") else (False, True)):
            L = torch.full((items, k, m), float("nan"), dtype=torch.float64, device=eng.device)
            nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device) if with_norms else None
            eng.latent_batch_index(d_vt if tiled else d_vs, n, idx, d_idx, L, nsq, tiled=tiled)
            assert eng.last_latent_kernel == ("full" if kind.endswith("full") else "index")
            np.testing.assert_allclose(L.cpu().numpy(), want, rtol=1e-11, atol=1e-11 * np.abs(want).max())
            if with_norms:
                np.testing.assert_allclose(nsq.cpu().numpy(), (vs ** 2).sum(-1), rtol=1e-12)


def test_latent_by_index_refuses_per_item():
    """An item with more different rows than the caller's bound, or with a row outside X, comes back as NaN; the
    other items of the launch are served."""
    import ctypes
    import torch
    from plspy_amd import _lib
    from plspy_amd.engine import ProjectionEngine, _ptr, _stream
    n, k, p, items, m = 64, 8, 333, 4, 64
    rs = np.random.RandomState(3)
    X = rs.randn(n, p)
    eng = ProjectionEngine(X)
    vs = rs.randn(items, k, p)
    idx = np.tile(np.arange(m, dtype=np.int32) % 40, (items, 1))         # 40 different rows
    idx[1] = np.arange(m)                                                # 64 different rows: over the bound
    idx[2, 5] = n                                                        # outside X
    d_vs, d_idx = eng.dev(vs), eng.dev(idx, torch.int32)
    L = torch.zeros((items, k, m), dtype=torch.float64, device=eng.device)
    lib = eng.lib
    need = lib.plsr_latent_index_workspace_bytes(n, k, items, p, m, 48, 0)
    assert need > 0
    work = torch.empty(need // 8, dtype=torch.float64, device=eng.device)
    _lib.check(lib.plsr_latent_index(_ptr(eng._xb(n)), p, n, _ptr(d_vs), d_vs.stride(1), 0, items, k,
                                     _ptr(d_idx), m, 48, None, 0, 0, None, 0, _ptr(L), None, _ptr(work), need,
                                     _stream()), "plsr_latent_index")
    got = L.cpu().numpy()
    assert np.isnan(got[1]).all() and np.isnan(got[2]).all()
    for b in (0, 3):
        np.testing.assert_allclose(got[b], vs[b] @ X[idx[b]].T, rtol=1e-11, atol=1e-11)
    assert lib.plsr_latent_index_workspace_bytes(129, k, items, p, m, 48, 0) == 0
    assert lib.plsr_latent_index_workspace_bytes(n, k, items, p, m, n + 1, 0) == 0
    assert lib.plsr_latent_index_workspace_bytes(n, k, items, p, m, 48, 17) == 0


@pytest.mark.parametrize("shape", [
    # n, k, p, items, m, rows of the item's own block, which of them
    (120, 38, 2051, 5, 80, 38, [0, 1, 2, 19, 20, 21]), (120, 38, 640, 3, 80, 38, [5]), (64, 16, 333, 4, 64, 20, list(range(16))),
    (120, 48, 1000, 3, 120, 48, [47, 0, 13]), (128, 16, 515, 2, 128, 9, [8, 7, 6, 5]),
])
def test_latent_by_index_with_own_rows(shape):
    """K5i with the extra tile of the item's own rows (the multiblock's raw task rows): the columns past m are
    VS_b . T_b[row_t]; row tiles + own tile within a wave's capacity, past it (second launch: the own tile alone, or
    with the last row tiles), a single own row, sixteen of them; the other columns unchanged."""
    import torch
    from plspy_amd.engine import ProjectionEngine
    n, k, p, items, m, trows, pick = shape
    rs = np.random.RandomState(n + k + p)
    X = rs.randn(n, p)
    eng = ProjectionEngine(X)
    vs = rs.randn(items, k, p)
    T = rs.randn(items, trows, p)
    idx = (np.stack([rs.permutation(n)[:m] for _ in range(items)]) if n == 128 else rs.randint(0, n, size=(items, m))).astype(np.int32)
    d_vs, d_idx, d_T = eng.dev(vs), eng.dev(idx, torch.int32), eng.dev(T)
    want = np.concatenate((np.einsum("bjv,biv->bji", vs, X[idx]), np.einsum("bjv,btv->bjt", vs, T[:, pick])), axis=2)
    for with_norms in (False, True):
        L = torch.full((items, k, m + len(pick)), float("nan"), dtype=torch.float64, device=eng.device)
        nsq = torch.empty((items, k), dtype=torch.float64, device=eng.device) if with_norms else None
        eng.latent_batch_index(d_vs, n, idx, d_idx, L, nsq, own=(d_T, pick))
        assert eng.last_latent_kernel == "index"
        np.testing.assert_allclose(L.cpu().numpy(), want, rtol=1e-11, atol=1e-11 * np.abs(want).max())
        if with_norms:
            np.testing.assert_allclose(nsq.cpu().numpy(), (vs ** 2).sum(-1), rtol=1e-12)

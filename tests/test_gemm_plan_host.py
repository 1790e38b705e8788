"""Host-only checks of the persistent GEMM's planner (grove_gemm_make_plan / grove_gemm_plan_image run without a device): the work list
of a temporal-tap-skipping Conv3d shape (grove_gemm_params.a_frame_rows / a_frames, round 4) covers every output tile's VALID K range
exactly once — first-frame tiles without the first tap group, last-frame tiles without the last — and a plain shape's image is still
exactly what grove_gemm_work_list describes."""
import ctypes as C

import numpy as np
import pytest


def _plan(M, N, K, taps=1, frames=(0, 0)):
    from grove_amd import _lib
    lib = _lib.lib()
    p = _lib.GemmParams()
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = K // taps, K, N
    p.batch1 = p.batch2 = 1
    p.a_taps, p.alpha = taps, 1.0
    p.A = p.B = p.C = C.c_void_p(0x10000)          # never dereferenced: plan and image are host arithmetic
    p.a_idx = C.c_void_p(0x10000) if taps > 1 else None
    p.a_frame_rows, p.a_frames = frames
    pl = _lib.GemmPlan()
    _lib.check(lib.grove_gemm_make_plan(C.byref(p), C.byref(pl)), "grove_gemm_make_plan")
    buf = np.zeros(int(pl.image_bytes), dtype=np.uint8)
    _lib.check(lib.grove_gemm_plan_image(C.byref(p), C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes)), "grove_gemm_plan_image")
    return pl, buf.view(np.int32).reshape(-1, 4)


def _coverage(pl, t):
    """{(m0, n0): sorted [(k0, k1, part)]} and the fix-up list of an image."""
    G = pl.grid
    n_fix = pl.scratch_bytes // (256 * 1024) // max(1, 1) if pl.stream_k else 0
    rows = (t.shape[0] - 0) // G
    head = t[:G]
    cov = {}
    for w in range(G):
        NT, nseg = int(head[w, 0]), int(head[w, 1])
        tot = 0
        for sgi in range(nseg):
            m0, n0, kk, part = (int(x) for x in t[(1 + sgi) * G + w])
            k0, k1 = kk & 0xffff, (kk >> 16) & 0xffff
            assert k1 > k0, "empty segment"
            cov.setdefault((m0, n0), []).append((k0, k1, part))
            tot += k1 - k0
        assert tot == NT
    return cov


@pytest.mark.parametrize("frames,T", [(32, 8), (16, 8), (64, 8), (24, 4)])
def test_tap_skipping_plan_covers_every_valid_k_tile_once(frames, T):
    HW, C_, taps = 1024, 1280, 27
    M, N, K = frames * HW, C_, taps * C_
    pl, t = _plan(M, N, K, taps, (HW, T))
    assert pl.bm == 256 and pl.k_tiles == K // 64
    nk = pl.k_tiles
    cov = _coverage(pl, t)
    assert len(cov) == pl.tiles_m * pl.tiles_n
    saved = 0
    split_tiles = 0
    for (m0, n0), segs in cov.items():
        tf = (m0 // HW) % T
        ka = nk // 3 if tf == 0 else 0
        kb = nk - nk // 3 if tf == T - 1 else nk
        segs = sorted(segs)
        assert segs[0][0] == ka and segs[-1][1] == kb, ((m0, n0), segs, ka, kb)
        for a, b in zip(segs, segs[1:]):
            assert a[1] == b[0]                      # contiguous, no overlap
        if len(segs) == 1:
            assert segs[0][2] == 0                   # a whole tile runs the epilogue itself
        else:
            split_tiles += 1
            parts = [s_[2] for s_ in segs]
            assert all(x > 0 for x in parts) and parts == list(range(parts[0], parts[0] + len(parts)))  # consecutive scratch slots in K order
        saved += nk - (kb - ka)
    assert saved == (2 * frames // T) * (HW // 256) * pl.tiles_n * (nk // 3)   # two frames per group lose a third each
    # the promise off: everything runs the whole range
    pl0, t0 = _plan(M, N, K, taps, (0, 0))
    assert all(sorted(v)[0][0] == 0 and sorted(v)[-1][1] == nk for v in _coverage(pl0, t0).values())
    assert pl0.key != pl.key
    # the longest block stream got shorter (that is the launch's duration)
    assert int(t[:pl.grid, 0].max()) < int(t0[:pl0.grid, 0].max())


def test_plain_shape_image_is_what_grove_gemm_work_list_reports():
    from grove_amd import _lib
    lib = _lib.lib()
    for (M, N, K) in [(32768, 1280, 5120), (2812, 12288, 4096), (18464, 1024, 1024)]:
        pl, t = _plan(M, N, K)
        cap = t.size + 64
        lst = np.zeros(cap, dtype=np.int32)
        fix = np.zeros(4096, dtype=np.int32)
        nf, kpp = C.c_int(0), C.c_int(0)
        rows = lib.grove_gemm_work_list(pl.bm, pl.tiles_m, pl.tiles_n, pl.k_tiles, 256, 1, lst.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(cap),
                                        fix.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(4096), C.byref(nf), C.byref(kpp))
        assert rows > 0 and kpp.value == pl.stream_k
        n = rows * pl.grid * 4
        assert np.array_equal(lst[:n], t.reshape(-1)[:n])
        assert np.array_equal(fix[:nf.value * 4], t.reshape(-1)[n:n + nf.value * 4])

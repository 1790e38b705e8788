"""The C-ABI library loads on a CPU-only box and exports every symbol include/grove_hip.h declares;
the ctypes mirror of every params struct has the size the library was compiled with."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from grove_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "grove_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(grove_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    from grove_amd import _lib
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libgrove_hip.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == declared, set(_lib.SYMBOLS) ^ set(declared)


def test_struct_sizes_match(lib):
    from grove_amd import _lib
    for name, st in _lib.STRUCTS.items():
        assert lib.grove_sizeof(name.encode()) == ctypes.sizeof(st), name


def test_errors_are_loud(lib):
    from grove_amd import _lib
    p = _lib.GemmParams()
    p.M, p.N, p.K = 4, 4, 7  # K not a multiple of 32 -> rejected before any launch
    rc = lib.grove_gemm_bf16(ctypes.byref(p), None)
    assert rc == -1 and "multiple of 32" in _lib.last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "grove_gemm_bf16")


def test_ops_refuse_cpu_tensors():
    import torch
    from grove_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(32, 32, dtype=torch.bfloat16), torch.zeros(32, 32, dtype=torch.bfloat16))


@pytest.mark.parametrize("bm,tiles_m,tiles_n,nk,mode", [
    (256, 128, 5, 80, 1),    # SAM fc2 / dgrad shapes: 640 tiles = 2 rounds + 128 -> halves
    (256, 128, 5, 20, 1),    # the same tiles with K = 1280: the split does not pay, whole tiles
    (256, 128, 5, 20, 2),    # ... forced
    (256, 79, 4, 40, 1),     # one round + 60 tiles -> four parts
    (192, 15, 16, 64, 1),    # fewer tiles than CUs (LLaMA o-proj): one partial round, never split
    (256, 11, 86, 64, 1),    # 946 tiles: tail of 178 > half a round, whole tiles
    (256, 3, 7, 5, 2),       # tiny
    (192, 97, 4, 64, 0),     # stream-K off
    (256, 73, 4, 64, 2),     # one round + 36 tiles -> four parts of 16
])
def test_gemm_work_list_covers_every_k_tile_once(lib, bm, tiles_m, tiles_n, nk, mode):
    """The host-made work list of the persistent GEMM kernels (grove_gemm_work_list: no device needed): every (output tile, K tile)
    appears in exactly one segment; whole tiles run K tiles 0..nk-1 with an epilogue (part 0); a split tile's parts tile [0, nk) in K
    order on consecutive workspace slots, one part per block, and are listed once in the fix-up list; a block's header carries the
    length of its stream; blocks of a round take neighbouring tiles."""
    import numpy as np
    G = 256
    tiles = tiles_m * tiles_n
    rows_cap = tiles // G + 4
    lst = np.zeros((rows_cap, G, 4), dtype=np.int32)
    fix = np.zeros((G, 4), dtype=np.int32)
    nf, S = ctypes.c_int(0), ctypes.c_int(0)
    rows = lib.grove_gemm_work_list(bm, tiles_m, tiles_n, nk, G, mode, lst.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(lst.size),
                                    fix.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(fix.size), ctypes.byref(nf), ctypes.byref(S))
    assert rows > 0
    grid = G if S.value else min(tiles, G)
    lst = lst.reshape(-1)[:rows * grid * 4].reshape(rows, grid, 4)
    cover = np.zeros((tiles_m, tiles_n, nk), dtype=np.int32)
    parts = {}
    for w in range(grid):
        NT, nseg = int(lst[0, w, 0]), int(lst[0, w, 1])
        assert nseg <= rows - 1
        total, split_seen = 0, 0
        for i in range(nseg):
            m0, n0, kk, part = (int(v) for v in lst[1 + i, w])
            k0, k1 = kk & 0xffff, (kk >> 16) & 0xffff
            assert m0 % bm == 0 and n0 % 256 == 0 and 0 <= k0 < k1 <= nk
            cover[m0 // bm, n0 // 256, k0:k1] += 1
            total += k1 - k0
            if part == 0:
                assert (k0, k1) == (0, nk), "a segment with an epilogue runs the whole K range"
            else:
                split_seen += 1
                assert i == nseg - 1, "a block's part of a split tile comes after its whole tiles"
                parts.setdefault((m0, n0), []).append((part - 1, k0, k1))
        assert total == NT and split_seen <= 1
    assert (cover == 1).all(), "every K tile of every output tile exactly once"
    if mode == 0 or S.value == 0:
        assert not parts and nf.value == 0
    else:
        assert nf.value == len(parts) == tiles % G and tiles // G >= 1 and tiles % G <= G // 2
        for j in range(nf.value):
            m0, n0, s0, np_ = (int(v) for v in fix[j])
            ps = sorted(parts[(m0, n0)])
            assert [p[0] for p in ps] == list(range(s0, s0 + np_)), "consecutive slots in K order"
            assert ps[0][1] == 0 and ps[-1][2] == nk and all(a[2] == b[1] for a, b in zip(ps, ps[1:]))
            assert max(p[2] - p[1] for p in ps) == S.value and 2 <= np_ <= 4
        assert len({s for v in parts.values() for s, _, _ in v}) == sum(len(v) for v in parts.values()) <= G
    if (bm, tiles_m, tiles_n, nk, mode) == (256, 128, 5, 80, 1):
        assert S.value == 40 and nf.value == 128
    if (bm, tiles_m, tiles_n, nk, mode) == (256, 128, 5, 20, 1):
        assert S.value == 0

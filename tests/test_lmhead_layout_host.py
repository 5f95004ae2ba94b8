"""Index arithmetic of the MFMA GEMM core (pipelinerl_amd/csrc/prl_lmhead_layout.h, the header the
kernels in prl_lmhead.hip include) compiled for the host: tile raster, the swizzled LDS image that
global_load_lds builds, the fragment reads and the accumulator map, emulated lane by lane.  Runs
without a GPU; the kernels themselves are covered by tests/test_gpu_lmhead_fused.py."""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
U16 = ctypes.POINTER(ctypes.c_uint16)
U8 = ctypes.POINTER(ctypes.c_uint8)


@pytest.fixture(scope="module")
def lmh(tmp_path_factory):
    out = tmp_path_factory.mktemp("harness") / "liblmhlayout.so"
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-o", str(out), str(ROOT / "tests" / "harness" / "lmhead_layout_host.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.lmh_tile_coords.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)] * 2
    lib.lmh_stage.argtypes = [U16, U8, ctypes.c_int, ctypes.c_int]
    lib.lmh_fragment.argtypes = [U8, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, U16, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.lmh_frag_byte.argtypes = [ctypes.c_int] * 4
    lib.lmh_frag_byte.restype = ctypes.c_int
    lib.lmh_stage32.argtypes = [U16, U8, ctypes.c_int, ctypes.c_int]
    lib.lmh_fragment32.argtypes = lib.lmh_fragment.argtypes
    lib.lmh_frag_byte32.argtypes = [ctypes.c_int] * 4
    lib.lmh_frag_byte32.restype = ctypes.c_int
    lib.lmh_emulate_tile.argtypes = [U16, U16, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int]
    lib.lmh_stage_tr.argtypes = [U16, U8, ctypes.c_int]
    lib.lmh_tr_read.argtypes = [U8, ctypes.POINTER(ctypes.c_int), U16]
    for name, n in (("lmh_tr_frag_byte", 5), ("lmh_tr_frag_entry", 3), ("lmh_tr_frag_token0", 3)):
        getattr(lib, name).argtypes = [ctypes.c_int] * n
        getattr(lib, name).restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("mt,nt", [(64, 8), (64, 1188), (1, 1), (3, 5), (8, 28), (16, 28), (1188, 28), (7, 3), (9, 1), (13, 17)])
def test_tile_raster_is_a_bijection_with_xcd_locality(lmh, mt, nt):
    seen = np.zeros((mt, nt), dtype=np.int32)
    tm, tn = ctypes.c_int(), ctypes.c_int()
    per_xcd: dict[int, list] = {}
    for bid in range(mt * nt):
        lmh.lmh_tile_coords(bid, mt, nt, ctypes.byref(tm), ctypes.byref(tn))
        assert 0 <= tm.value < mt and 0 <= tn.value < nt
        seen[tm.value, tn.value] += 1
        per_xcd.setdefault(bid % 8, []).append((tm.value, tn.value))
    assert (seen == 1).all()
    if mt % 8 == 0 and mt * nt >= 512 and nt >= 8:
        # the first 64 tiles an XCD receives (what its 32 CUs run at once) span at most 8 row tiles and ~8 column tiles
        for tiles in per_xcd.values():
            first = tiles[:64]
            assert len({t[0] for t in first}) <= 8 and len({t[1] for t in first}) <= 9


@pytest.mark.parametrize("rows,nthreads", [(128, 256), (256, 512), (128, 512)])  # 256 rows / 512 threads is also the B tile of the 256 x 256 shape
def test_staged_image_matches_the_fragment_reads(lmh, rows, nthreads):
    """Every (row, k) of a tile reaches the lane the MFMA operand layout assigns it to - for the A tile
    of both workgroup shapes (128 rows / 256 threads, 256 rows / 512 threads) and the 128-row B tile
    staged by 512 threads."""
    src = (np.arange(rows)[:, None] * 64 + np.arange(64)[None, :]).astype(np.uint16)  # value = row * 64 + k
    lds = np.zeros(rows * 128, dtype=np.uint8)
    lmh.lmh_stage(src.ctypes.data_as(U16), lds.ctypes.data_as(U8), rows, nthreads)
    assert sorted(np.frombuffer(lds.tobytes(), dtype=np.uint16).tolist()) == sorted(src.reshape(-1).tolist())
    out = (ctypes.c_uint16 * 8)()
    row, k0 = ctypes.c_int(), ctypes.c_int()
    covered = set()
    for w in range(rows // 64):
        for i in range(2):
            for ks in range(4):
                for lane in range(64):
                    lmh.lmh_fragment(lds.ctypes.data_as(U8), lane, w * 64, i, ks, out, ctypes.byref(row), ctypes.byref(k0))
                    assert row.value == w * 64 + i * 32 + (lane & 31)
                    assert list(out) == [row.value * 64 + k0.value + e for e in range(8)]
                    covered.update((row.value, k0.value + e) for e in range(8))
    assert len(covered) == rows * 64


# lane groups ds_read_b128 is served in, one LDS cycle each when conflict-free (MI355X_MICROARCH.md, LDS table)
B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def test_fragment_reads_are_bank_conflict_free(lmh):
    """ds_read_b128 is served per lane group over a 256-byte bank row (64 banks x 4 B): the 16 lanes of a
    group must hit 16 distinct 16-byte slots."""
    for row0 in (0, 64, 128, 192):
        for i in range(4):  # up to four 32-row tiles per wave (B side of the 256-column shape)
            for ks in range(4):
                for group in B128_GROUPS:
                    slots = {(lmh.lmh_frag_byte(l, row0, i, ks) % 256) // 16 for l in group}
                    assert len(slots) == 16


@pytest.mark.parametrize("wm_count,bn", [(2, 128), (4, 128), (4, 256)])
def test_emulated_tile_product_is_not_transposed(lmh, wm_count, bn):
    """C = A B^T of one (64 wm_count) x bn x 64 tile through staging, fragments and the MFMA lane maps,
    with an asymmetric B so that a row/column swap anywhere in the chain shows."""
    rng = np.random.default_rng(0)
    bm = 64 * wm_count
    a = rng.integers(0, 7, size=(bm, 64)).astype(np.uint16)
    b = rng.integers(0, 5, size=(bn, 64)).astype(np.uint16)
    b[:, 0] += np.arange(bn, dtype=np.uint16)  # rows of B differ systematically
    c = np.zeros((bm, bn), dtype=np.float64)
    lmh.lmh_emulate_tile(a.ctypes.data_as(U16), b.ctypes.data_as(U16), c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), wm_count, bn)
    want = a.astype(np.float64) @ b.astype(np.float64).T
    assert np.array_equal(c, want)


def test_32_deep_layout_of_the_dual_plane_core(lmh):
    """256 rows x 32 contraction elements staged by 512 threads: every (row, k) reaches the lane the MFMA
    operand layout assigns it to, and the fragment reads are bank-conflict free for the real lane groups."""
    rows, nthreads = 256, 512
    src = (np.arange(rows)[:, None] * 32 + np.arange(32)[None, :]).astype(np.uint16)
    lds = np.zeros(rows * 64, dtype=np.uint8)
    lmh.lmh_stage32(src.ctypes.data_as(U16), lds.ctypes.data_as(U8), rows, nthreads)
    assert sorted(np.frombuffer(lds.tobytes(), dtype=np.uint16).tolist()) == sorted(src.reshape(-1).tolist())
    out = (ctypes.c_uint16 * 8)()
    row, k0 = ctypes.c_int(), ctypes.c_int()
    covered = set()
    for row0 in (0, 64, 128, 192):
        for i in range(2):
            for ks in range(2):
                for lane in range(64):
                    lmh.lmh_fragment32(lds.ctypes.data_as(U8), lane, row0, i, ks, out, ctypes.byref(row), ctypes.byref(k0))
                    assert row.value == row0 + i * 32 + (lane & 31)
                    assert list(out) == [row.value * 32 + k0.value + e for e in range(8)]
                    covered.update((row.value, k0.value + e) for e in range(8))
    assert len(covered) == rows * 32
    for row0 in (0, 128):
        for i in range(4):
            for ks in range(2):
                for group in B128_GROUPS:
                    slots = {(lmh.lmh_frag_byte32(l, row0, i, ks) % 256) // 16 for l in group}
                    assert len(slots) == 16


def test_transposing_reads_feed_the_mfma_from_a_row_major_tile(lmh):
    """d W from the ROW-MAJOR d-logits planes: the staged [32 tokens][256 entries] image and the per-lane addresses of
    `ds_read_b64_tr_b16` (semantics as measured on the device, profiles/r03b_mx_probe.txt) hand lane l of wave row w,
    tile i, sub-step ks exactly the operand the MFMA expects - entry w + 32 i + (l & 31), tokens 16 ks + 8 (l >> 5) + 0..7 -
    and the 32 lanes of one LDS pass touch 64 distinct banks."""
    src = (np.arange(32)[:, None] * 256 + np.arange(256)[None, :]).astype(np.uint16)  # value = token * 256 + entry
    lds = np.zeros(32 * 512, dtype=np.uint8)
    lmh.lmh_stage_tr(src.ctypes.data_as(U16), lds.ctypes.data_as(U8), 512)
    assert sorted(np.frombuffer(lds.tobytes(), dtype=np.uint16).tolist()) == sorted(src.ravel().tolist())  # a permutation of the tile
    for w in (0, 64, 128, 192):
        for i in (0, 1):
            for ks in (0, 1):
                frag = np.zeros((64, 8), dtype=np.int64)
                for r in (0, 1):
                    addr = np.array([lmh.lmh_tr_frag_byte(l, w, i, ks, r) for l in range(64)], dtype=np.int32)
                    assert (addr % 8 == 0).all() and addr.max() + 8 <= lds.size
                    for half in (0, 1):  # one LDS pass = 32 lanes x 8 bytes: every 4-byte bank exactly once
                        banks = np.concatenate([((addr[32 * half:32 * half + 32] + d) // 4) % 64 for d in (0, 4)])
                        assert len(set(banks.tolist())) == 64, (w, i, ks, r, half)
                    out = np.zeros(64 * 4, dtype=np.uint16)
                    lmh.lmh_tr_read(lds.ctypes.data_as(U8), addr.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), out.ctypes.data_as(U16))
                    frag[:, 4 * r:4 * r + 4] = out.reshape(64, 4)
                for l in range(64):
                    entry = lmh.lmh_tr_frag_entry(l, w, i)
                    assert entry == w + 32 * i + (l & 31)
                    want = [(16 * ks + 8 * (l >> 5) + e) * 256 + entry for e in range(8)]
                    assert frag[l].tolist() == want, (w, i, ks, l)
                    assert lmh.lmh_tr_frag_token0(l, ks, 1) == 16 * ks + 8 * (l >> 5) + 4

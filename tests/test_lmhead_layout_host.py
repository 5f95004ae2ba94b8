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
    lib.lmh_stage.argtypes = [U16, U8]
    lib.lmh_fragment.argtypes = [U8, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, U16, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.lmh_frag_byte.argtypes = [ctypes.c_int] * 4
    lib.lmh_frag_byte.restype = ctypes.c_int
    lib.lmh_emulate_tile.argtypes = [U16, U16, ctypes.POINTER(ctypes.c_double)]
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


def test_staged_image_matches_the_fragment_reads(lmh):
    """Every (row, k) of a tile reaches the lane the MFMA operand layout assigns it to."""
    src = (np.arange(128)[:, None] * 64 + np.arange(64)[None, :]).astype(np.uint16)  # value = row * 64 + k
    lds = np.zeros(16384, dtype=np.uint8)
    lmh.lmh_stage(src.ctypes.data_as(U16), lds.ctypes.data_as(U8))
    assert sorted(np.frombuffer(lds.tobytes(), dtype=np.uint16).tolist()) == sorted(src.reshape(-1).tolist())
    out = (ctypes.c_uint16 * 8)()
    row, k0 = ctypes.c_int(), ctypes.c_int()
    covered = set()
    for w in range(2):
        for i in range(4):
            for ks in range(2):
                for lane in range(64):
                    lmh.lmh_fragment(lds.ctypes.data_as(U8), lane, w, i, ks, out, ctypes.byref(row), ctypes.byref(k0))
                    assert row.value == w * 64 + i * 16 + (lane & 15)
                    assert list(out) == [row.value * 64 + k0.value + e for e in range(8)]
                    covered.update((row.value, k0.value + e) for e in range(8))
    assert len(covered) == 128 * 64


def test_fragment_reads_are_bank_conflict_free(lmh):
    """ds_read_b128 is served per 16-lane group over a 256-byte bank row (64 banks x 4 B): the 16 lanes
    of a group must hit 16 distinct 16-byte slots."""
    for w in range(2):
        for i in range(4):
            for ks in range(2):
                for group in range(4):
                    slots = {(lmh.lmh_frag_byte(group * 16 + l, w, i, ks) % 256) // 16 for l in range(16)}
                    assert len(slots) == 16


def test_emulated_tile_product_is_not_transposed(lmh):
    """C = A B^T of one 128 x 128 x 64 tile through staging, fragments and the MFMA lane maps, with an
    asymmetric B so that a row/column swap anywhere in the chain shows."""
    rng = np.random.default_rng(0)
    a = rng.integers(0, 7, size=(128, 64)).astype(np.uint16)
    b = rng.integers(0, 5, size=(128, 64)).astype(np.uint16)
    b[:, 0] += np.arange(128, dtype=np.uint16)  # rows of B differ systematically
    c = np.zeros((128, 128), dtype=np.float64)
    lmh.lmh_emulate_tile(a.ctypes.data_as(U16), b.ctypes.data_as(U16), c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    want = a.astype(np.float64) @ b.astype(np.float64).T
    assert np.array_equal(c, want)
    assert not np.array_equal(want, want.T)

"""Error paths of the C ABI on a real device: the reference's hard limits (29-bit original positions,
src/stranditerator.cpp:19-27; total input <= 2^30, src/common.h:52 + src/sibelia.cpp:232-235; int32 suffix ranks of the long-k path,
src/vertexenumeration.cpp:288-300) come back as SBL_ERR_TOO_LARGE, an allocation the device cannot satisfy as SBL_ERR_OOM -- as codes,
with a message, leaving the context usable -- and never as a crash or a hang."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SBL_ERR_OOM, SBL_ERR_TOO_LARGE = 3, 5


def _ctx():
    from sibelia_amd.api import load_library
    L = load_library()
    h = C.c_void_p()
    assert L.sbl_create(C.byref(h), 0) == 0
    return L, h


def _load(L, h, bufs, lens):
    n = len(bufs)
    arr = (C.c_char_p * n)(*bufs)
    ln = (C.c_uint64 * n)(*lens)
    return L.sbl_load(h, n, arr, ln)


def test_chromosome_of_2_pow_29_is_too_large_and_the_context_survives():
    # the limits are checked on the LENGTHS before a single byte is read: no 512 MB buffer needed
    L, h = _ctx()
    try:
        assert _load(L, h, [b"ACGT"], [1 << 29]) == SBL_ERR_TOO_LARGE
        assert b"2^29" in L.sbl_last_error(h)
        assert _load(L, h, [b"ACGT", b"ACGT", b"ACGT"], [(1 << 29) - 1, (1 << 29) - 1, 3]) == SBL_ERR_TOO_LARGE      # sum = 2^30 + 1
        assert b"2^30" in L.sbl_last_error(h)
        # ... and the same context still loads and enumerates a legal input
        assert _load(L, h, [b"ACGTTGCAAGGCTTACGGATCC", b"ACGTTGCAAGGCTAACGGATCC"], [22, 22]) == 0
        bc = C.c_uint32(); p, q = C.c_void_p(), C.c_void_p(); n, m = C.c_uint64(), C.c_uint64()
        assert L.sbl_enumerate(h, 5, C.byref(bc), C.byref(p), C.byref(n), C.byref(q), C.byref(m)) == 0 and bc.value > 0
    finally:
        L.sbl_destroy(h)


def test_fasta_loader_enforces_the_same_limits(tmp_path):
    # sbl_load_fasta parses on the device and applies the limits to what it parsed: one record of exactly 2^29 bases
    L, h = _ctx()
    try:
        fa = tmp_path / "big.fa"
        line = b"ACGT" * 16384 + b"\n"                      # 65 536 bases per line
        with open(fa, "wb") as f:
            f.write(b">r0\n")
            f.write(line * ((1 << 29) // 65536))
        assert L.sbl_load_fasta(h, str(fa).encode()) == SBL_ERR_TOO_LARGE
        assert b"2^29" in L.sbl_last_error(h)
    finally:
        L.sbl_destroy(h)


def test_long_k_rejects_more_than_2_pow_31_suffixes():
    # total input = 2^30 exactly is legal (common.h:52) but 2 x 2^30 + separators does not fit the reference's int32 suffix array
    # (vertexenumeration.cpp:288-300), nor the 32-bit ranks here: a clean code, before any workspace is allocated
    L, h = _ctx()
    try:
        a = np.full((1 << 29) - 1, ord("A"), np.uint8).tobytes()
        rest = (1 << 30) - 2 * ((1 << 29) - 1)
        assert _load(L, h, [a, a, b"C" * rest], [len(a), len(a), rest]) == 0
        del a
        bc = C.c_uint32(); p, q = C.c_void_p(), C.c_void_p(); n, m = C.c_uint64(), C.c_uint64()
        assert L.sbl_enumerate(h, 100, C.byref(bc), C.byref(p), C.byref(n), C.byref(q), C.byref(m)) == SBL_ERR_TOO_LARGE
        assert b"suffix" in L.sbl_last_error(h)
    finally:
        L.sbl_destroy(h)


def test_device_out_of_memory_is_a_code_not_a_crash(monkeypatch):
    """SBL_TEST_ALLOC_LIMIT_MB caps what one context may hold on the device (DevBuf::ensure): the stage fails with SBL_ERR_OOM and a
    message naming the allocation, the input state is untouched, and the same stage succeeds once the cap is lifted."""
    from sibelia_amd import BlockFinder, SibeliaError, workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=200_000, n=4, seed=9, inv_min=2000, inv_max=8000)
    bf, orc = BlockFinder(seqs, device=0), Oracle(seqs)
    try:
        monkeypatch.setenv("SBL_TEST_ALLOC_LIMIT_MB", "64")
        with pytest.raises(SibeliaError) as e:
            bf.simplify_stage(25, 150, 4)
        assert "out of memory" in str(e.value).lower() or "oom" in str(e.value).lower()
        monkeypatch.delenv("SBL_TEST_ALLOC_LIMIT_MB")
        assert bf.simplify_stage(25, 150, 4) == orc.simplify_stage(25, 150, 4)
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    finally:
        bf.close()


def test_long_k_preflight_reports_missing_memory(monkeypatch):
    """the long-k path sizes its workspaces (~56 B per suffix) up front and compares with what the device has free: a clear
    SBL_ERR_OOM before the first allocation instead of a failure half way through (SBL_TEST_FREE_MEM_MB pretends less is free)"""
    from sibelia_amd import BlockFinder, SibeliaError, workloads as W
    seqs = W.longk_case(2_000_000, 4)
    bf = BlockFinder(seqs, device=0)
    try:
        monkeypatch.setenv("SBL_TEST_FREE_MEM_MB", "32")
        with pytest.raises(SibeliaError) as e:
            bf.enumerate(5000)
        assert "long-k" in str(e.value) and "free" in str(e.value)
        monkeypatch.delenv("SBL_TEST_FREE_MEM_MB")
        assert bf.enumerate(5000)[0] > 0
    finally:
        bf.close()


def test_a_rank_failing_inside_the_simplification_releases_its_peers(monkeypatch):
    """the read-only phases of the simplification are collectives too (verdict bytes all-gathered per snapshot and per probe): a rank
    that leaves the stage with an error anywhere must not leave its peers waiting there (SBL_TEST_FAIL_SIMPLIFY_RANK: that rank throws
    SBL_ERR_OOM between the enumeration and the first snapshot)"""
    from sibelia_amd import SibeliaError, workloads as W
    from sibelia_amd.dist import LocalShardedFinder
    seqs = W.gen_strains(L0=120_000, n=4, seed=11, inv_min=2000, inv_max=8000)
    f = LocalShardedFinder(seqs, [0, 0, 0])
    try:
        monkeypatch.setenv("SBL_TEST_FAIL_SIMPLIFY_RANK", "2")
        with pytest.raises(SibeliaError):
            f.simplify_stage(25, 150, 4)
        monkeypatch.delenv("SBL_TEST_FAIL_SIMPLIFY_RANK")
    finally:
        f.close()


def test_a_failing_virtual_rank_releases_its_peers(monkeypatch):
    """sharded abort path: one of three virtual ranks fails with SBL_ERR_OOM inside the collective enumeration; its peers must
    come back with an error instead of waiting at the barrier for ever (SBL_TEST_FAIL_RANK: that rank throws SBL_ERR_OOM after the exchange)"""
    from sibelia_amd import SibeliaError, workloads as W
    from sibelia_amd.dist import LocalShardedFinder
    seqs = W.gen_strains(L0=200_000, n=4, seed=10, inv_min=2000, inv_max=8000)
    f = LocalShardedFinder(seqs, [0, 0, 0])
    try:
        monkeypatch.setenv("SBL_TEST_FAIL_RANK", "1")
        with pytest.raises(SibeliaError):
            f.enumerate(25)
        monkeypatch.delenv("SBL_TEST_FAIL_RANK")
    finally:
        f.close()

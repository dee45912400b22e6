"""The C-ABI library on a machine without a GPU: it loads, exports every symbol include/em2_lsh.h declares,
its host-side pieces agree with the oracle, and its device entry points fail loudly (no CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import ctypes
import pytest

from expressionmatrix2_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIBRARY_PATH):
        capi.build_library()
    return capi.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "em2_lsh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(em2_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = declared_functions()
    assert len(declared) >= 15
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIBRARY_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (em2_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(capi.SYMBOLS), sorted(set(declared) ^ set(capi.SYMBOLS))
    assert lib.em2_abi_version() == 1


def test_library_holds_gfx950_code(lib):
    blob = open(capi.LIBRARY_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"fsp4ScanKernel" in blob and b"projectionKernel" in blob


def test_product_library_has_no_measurement_knobs(lib):
    """EM2_MATRIX_DIAG / EM2_PROJECTION_DIAG switch parts of kernels off and give wrong SimilarPairs by design: they are
    compiled only into libem2lsh_diag.so (make diag, -DEM2_DIAG).  The product library must not even contain the names."""
    blob = open(capi.LIBRARY_PATH, "rb").read()
    assert b"EM2_MATRIX_DIAG" not in blob and b"EM2_PROJECTION_DIAG" not in blob
    makefile = open(os.path.join(ROOT, "expressionmatrix2_amd", "csrc", "Makefile")).read()
    assert "-DEM2_DIAG" in makefile and "libem2lsh_diag.so" in makefile


def test_product_library_reads_few_environment_variables(lib):
    """Test knobs that force rare paths at small sizes, and nothing else: at most 20 EM2_* names in the product library (round 4
    had 38, a third of them A/B partners of measurements long settled), every one of them in DESIGN.md's table."""
    blob = open(capi.LIBRARY_PATH, "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"EM2_[A-Z0-9_]+", blob)))
    assert len(names) <= 20, names
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for name in names:
        assert "`%s`" % name in design, name
    assert os.path.getsize(capi.LIBRARY_PATH) < 10 * 1024 * 1024


def test_generate_vectors_matches_oracle(lib, oracle):
    # (from 65 536 values on the engine runs 624 values at a time on the calling thread and the variates are formed behind it by
    # the other host threads, chunk by chunk of 65 536 pairs: sizes of one, three and sixteen chunks, an odd total among them)
    for genes, L, seed in [(7, 64, 231), (50, 128, 231), (33, 100, 5), (1, 1, 9), (64, 1024, 231), (300, 1024, 231), (257, 1001, 7),
                           (2000, 1024, 1), (1, 65537, 4294967295)]:
        a = capi.lsh_generate_vectors(genes, L, seed)
        b = oracle.generate_lsh_vectors(genes, L, seed)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
        if genes > 1:
            assert np.allclose((a * a).sum(axis=0), 1.0, atol=1e-12)


def test_similarity_table_matches_oracle(lib, oracle):
    for L in (1, 64, 128, 1000, 1024, 2048, 4096):
        assert np.array_equal(capi.similarity_table(L).view(np.uint64), oracle.similarity_table(L).view(np.uint64))


def test_murmur_matches_oracle(lib, oracle):
    for n in (0, 1, 7, 8, 9, 4000):
        data = (np.arange(n, dtype=np.uint64) * 2654435761 % 251).astype(np.uint8)
        assert capi.murmur_hash_64a(data) == oracle.murmur(data)


def test_device_paths_fail_loudly_without_gpu(lib):
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    sig = np.zeros((4, 2), dtype=np.uint64)
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.find_similar_pairs4(sig, 128, 3, 0.2)
    toc = np.zeros(5, dtype=np.uint64)
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.compute_signatures(toc, np.zeros(0, dtype=capi.COUNT_DTYPE), 3, np.zeros((3, 64)), 64)
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.find_similar_pairs5(sig, 128, 3, 0.2, 8)
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.find_similar_pairs7(sig, 128, 3, 0.2, [10, 8], 100, 12)
    pairs = np.zeros((4, 3), dtype=capi.PAIR_DTYPE)
    used = np.zeros(4, dtype=np.uint32)
    cells = np.arange(4, dtype=np.uint32)
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.cell_graph_edges(pairs, used, cells, cells, 0.2, 3)
    # the fused subset call has no numpy wrapper (the facade drives it through the matrix handle): raw ABI
    counts = np.zeros(0, dtype=capi.COUNT_DTYPE)
    genes = np.arange(3, dtype=np.uint32)
    vectors = np.zeros((64, 3))
    status = capi.load().em2_subset_find_similar_pairs4(
        toc.ctypes.data, counts.ctypes.data, 4, cells.ctypes.data, 4, genes.ctypes.data, 3, 3,
        vectors.ctypes.data, 64, None, 3, 0.2, pairs.ctypes.data, used.ctypes.data)
    assert status != 0
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.check(status)


def test_scan_form_query_knows_about_the_matrix_cores(lib, monkeypatch):
    # em2_dev_find_similar_pairs4_form_for: 129..2048-bit signatures take the symmetric form on the matrix cores (3) from
    # 32768 cells on (EM2_SCAN_MATRIX=2: every width up to 2048; 3: none above 1024), other widths the
    # v_xor/v_bcnt symmetric form (1) from 131072 cells on; a row shard is never symmetric.
    monkeypatch.delenv("EM2_SCAN_MODE", raising=False)
    monkeypatch.delenv("EM2_SCAN_MATRIX", raising=False)
    f = lib.em2_dev_find_similar_pairs4_form_for
    assert f(100000, 100000, 1024) == 3 and f(100000, 100000, 600) == 3
    assert f(20000, 20000, 1024) == 0
    assert f(100000, 100000, 512) == 3 and f(100000, 100000, 129) == 3
    assert f(100000, 100000, 128) == 0 and f(200000, 200000, 128) == 1 and f(200000, 200000, 4096) == 1
    assert f(100000, 100000, 2048) == 3 and f(100000, 100000, 1025) == 3 and f(20000, 20000, 2048) == 0
    monkeypatch.setenv("EM2_SCAN_MATRIX", "3")
    assert f(100000, 100000, 2048) == 0 and f(200000, 200000, 2048) == 1 and f(100000, 100000, 1024) == 3
    monkeypatch.setenv("EM2_SCAN_MATRIX", "2")
    assert f(100000, 100000, 64) == 3 and f(200000, 200000, 2048) == 3 and f(200000, 200000, 3000) == 1
    # a shard of the rows (never symmetric) takes the rows form on the matrix cores (4) from 2^31 (row, column) pairs on
    assert f(200000, 100000, 1024) == 4 and f(200000, 100000, 2048) == 4 and f(200000, 100000, 3000) == 0
    assert f(200000, 10000, 1024) == 0 and f(1000000, 125000, 1024) == 4 and f(1000000, 2048, 1024) == 0
    monkeypatch.setenv("EM2_SCAN_MATRIX", "0")
    assert f(100000, 100000, 1024) == 0 and f(200000, 200000, 1024) == 1 and f(200000, 100000, 1024) == 0
    monkeypatch.delenv("EM2_SCAN_MATRIX")
    monkeypatch.setenv("EM2_SCAN_MODE", "rows")               # (tests: the rows form for every launch it can serve)
    assert f(3000, 3000, 1024) == 4 and f(3000, 1, 1024) == 4 and f(100000, 100000, 1024) == 4 and f(3000, 3000, 64) == 0
    monkeypatch.setenv("EM2_SCAN_MODE", "persistent")
    assert f(200000, 100000, 1024) == 0
    monkeypatch.delenv("EM2_SCAN_MODE")
    w = lib.em2_dev_find_similar_pairs4_workspace
    w.restype = ctypes.c_size_t
    # (the rows form's workspace: fragments of every column, and of the rows once more when they are not all cells)
    assert w(200000, 100000, 1024, 100) - w(200000, 100000, 128, 100) > 200000 * 512 + 100000 * 512
    assert lib.em2_dev_find_similar_pairs4_form(200000, 200000) == 1

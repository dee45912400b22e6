"""ctypes binding of the C ABI declared in include/em2_lsh.h (libem2lsh.so, built for gfx950).

This module is plumbing only: it loads the shared library that holds the HIP kernels and exposes numpy /
device-pointer level wrappers.  There is no Python or CPU implementation of the path behind it: if the
library is missing or no GPU is visible the calls raise RuntimeError.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.environ.get("EM2_LIBRARY") or os.path.join(_HERE, "libem2lsh.so")   # override: A/B builds only
CSRC_DIR = os.path.join(_HERE, "csrc")

# std::pair<CellId,float> (src/SimilarPairs.hpp:53-56) and std::pair<GeneId,float> (src/ExpressionMatrixSubset.hpp:36)
PAIR_DTYPE = np.dtype([("cell", "<u4"), ("similarity", "<f4")])
COUNT_DTYPE = np.dtype([("gene", "<u4"), ("count", "<f4")])

EM2_OK = 0
EM2_ERROR_NO_DEVICE = 2
EM2_ERROR_UNSUPPORTED = 6

_lib = None

# name -> (restype, argtypes); every symbol include/em2_lsh.h declares.
_c = ctypes
SYMBOLS = {
    "em2_abi_version": (_c.c_int, []),
    "em2_last_error": (_c.c_char_p, []),
    "em2_lsh_generate_vectors": (_c.c_int, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_void_p]),
    "em2_lsh_similarity_table": (_c.c_int, [_c.c_uint32, _c.c_void_p]),
    "em2_murmur_hash_64a": (_c.c_uint64, [_c.c_void_p, _c.c_int, _c.c_uint64]),
    "em2_device_count": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "em2_set_device": (_c.c_int, [_c.c_int]),
    "em2_compute_signatures": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p,
                                          _c.c_uint32, _c.c_void_p]),
    "em2_find_similar_pairs4": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_double,
                                           _c.c_void_p, _c.c_void_p]),
    "em2_find_similar_pairs5": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_double,
                                           _c.c_uint32, _c.c_uint64, _c.c_void_p, _c.c_void_p]),
    "em2_dev_compute_signatures_workspace": (_c.c_size_t, [_c.c_uint32, _c.c_uint32]),
    "em2_dev_compute_signatures": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p,
                                              _c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_void_p, _c.c_size_t,
                                              _c.c_void_p]),
    "em2_dev_compute_signatures_tier": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_int, _c.c_void_p]),
    "em2_dev_vector_aux_bytes": (_c.c_size_t, [_c.c_uint32, _c.c_uint32]),
    "em2_dev_prepare_vectors": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p, _c.c_void_p]),
    "em2_dev_find_similar_pairs4_workspace": (_c.c_size_t, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "em2_dev_find_similar_pairs4": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                               _c.c_uint32, _c.c_double, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                               _c.c_size_t, _c.c_void_p]),
    "em2_find_similar_pairs7": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_double, _c.c_void_p,
                                           _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_void_p, _c.c_void_p]),
    "em2_dev_find_similar_pairs7": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                               _c.c_double, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                               _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "em2_matrix_find_similar_pairs7": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p, _c.c_char_p, _c.c_char_p,
                                                  _c.c_size_t, _c.c_double, _c.c_void_p, _c.c_uint32, _c.c_uint32,
                                                  _c.c_size_t]),
    "em2_dev_subset_workspace": (_c.c_size_t, [_c.c_uint32]),
    "em2_dev_subset_count": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_uint32,
                                        _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "em2_dev_subset_fill": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_uint32,
                                       _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "em2_subset_find_similar_pairs4": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_uint32,
                                                  _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p, _c.c_uint32,
                                                  _c.c_void_p, _c.c_uint32, _c.c_double, _c.c_void_p, _c.c_void_p]),
    "em2_cell_graph_label_propagation": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                    _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_void_p,
                                                    _c.c_void_p]),
    "em2_dev_cell_graph_label_propagation": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                        _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_void_p,
                                                        _c.c_void_p]),
    "em2_analyze_lsh": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p, _c.c_uint32, _c.c_void_p,
                                   _c.c_uint32, _c.c_double, _c.c_char_p, _c.c_char_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                   _c.c_void_p, _c.c_void_p]),
    "em2_matrix_analyze_lsh": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p, _c.c_size_t, _c.c_uint, _c.c_double,
                                          _c.c_char_p]),
    "em2_dist_find_similar_pairs4_workspace": (_c.c_size_t, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "em2_dist_find_similar_pairs4_form": (_c.c_int, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "em2_dist_find_similar_pairs4": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_double,
                                                _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                                _c.c_void_p]),
    "em2_dist_find_similar_pairs4_with": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_double,
                                                     _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                                     _c.c_void_p]),
    "em2_dev_find_similar_pairs4_form": (_c.c_int, [_c.c_uint32, _c.c_uint32]),
    "em2_dev_find_similar_pairs4_form_for": (_c.c_int, [_c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "em2_dev_find_similar_pairs4_last_launch": (_c.c_int, [_c.c_void_p, _c.c_uint32]),
    "em2_dev_find_similar_pairs5_last_launch": (_c.c_int, [_c.c_void_p, _c.c_uint32]),
    "em2_dev_release_scratch": (None, []),
    "em2_dev_fsp4_sharded_plan": (_c.c_int, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                             _c.c_void_p, _c.c_uint32]),
    "em2_dev_fsp4_sharded_phase": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_double,
                                              _c.c_uint32, _c.c_uint32, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                              _c.c_size_t, _c.c_uint64, _c.c_void_p]),
    "em2_dev_fsp4_sharded_status": (_c.c_int, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_void_p,
                                               _c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint32)]),
    "em2_dev_find_similar_pairs4_status": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p]),
    "em2_dev_find_similar_pairs5": (_c.c_int, [_c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                               _c.c_uint32, _c.c_double, _c.c_uint32, _c.c_uint64, _c.c_void_p,
                                               _c.c_void_p, _c.c_void_p]),
    "em2_matrix_open": (_c.c_int, [_c.c_char_p, _c.POINTER(_c.c_void_p)]),
    "em2_matrix_close": (None, [_c.c_void_p]),
    "em2_matrix_find_similar_pairs4": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p, _c.c_char_p, _c.c_size_t,
                                                  _c.c_double, _c.c_size_t, _c.c_uint]),
    "em2_matrix_compute_lsh_signatures": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p, _c.c_char_p,
                                                     _c.c_size_t, _c.c_uint]),
    "em2_matrix_find_similar_pairs5": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p, _c.c_char_p, _c.c_char_p,
                                                  _c.c_size_t, _c.c_double, _c.c_size_t, _c.c_size_t]),
    "em2_matrix_remove_similar_pairs": (_c.c_int, [_c.c_void_p, _c.c_char_p]),
    "em2_matrix_subset": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_char_p, _c.POINTER(_c.c_uint32),
                                     _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_uint64), _c.c_void_p, _c.c_void_p]),
    "em2_similar_pairs_write": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.c_char_p, _c.c_char_p, _c.c_size_t,
                                           _c.c_uint32, _c.c_void_p, _c.c_void_p]),
    "em2_similar_pairs_read": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.POINTER(_c.c_uint64),
                                          _c.POINTER(_c.c_uint64), _c.c_void_p, _c.c_void_p]),
    "em2_similar_pairs_info": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.POINTER(_c.c_uint64),
                                          _c.POINTER(_c.c_uint64), _c.c_char_p, _c.c_char_p]),
    "em2_matrix_cell_set": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.POINTER(_c.c_uint32), _c.c_void_p]),
    "em2_cell_graph_edges": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p,
                                        _c.c_void_p, _c.c_uint32, _c.c_double, _c.c_uint32, _c.c_void_p,
                                        _c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_uint64)]),
    "em2_dev_cell_graph_edges": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint32, _c.c_void_p,
                                            _c.c_void_p, _c.c_uint32, _c.c_double, _c.c_uint32, _c.c_void_p,
                                            _c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_uint64)]),
    "em2_lsh_write": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.c_uint64, _c.c_uint64, _c.c_void_p]),
    "em2_lsh_read": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64),
                                _c.c_void_p]),
    "em2_tool_create_directory": (_c.c_int, [_c.c_char_p, _c.c_uint32, _c.c_uint32, _c.c_void_p, _c.c_void_p]),
    "em2_tool_add_gene_set": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.c_void_p, _c.c_uint32]),
    "em2_tool_add_cell_set": (_c.c_int, [_c.c_char_p, _c.c_char_p, _c.c_void_p, _c.c_uint32]),
}


def build_library(verbose=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    result = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or result.returncode != 0:
        print(result.stdout)
        print(result.stderr)
    if result.returncode != 0:
        raise RuntimeError("building libem2lsh.so failed:\n" + result.stderr[-4000:])
    return LIBRARY_PATH


def _share_hip_runtime_with_torch():
    """torch wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process cannot both see the GPU,
    and the only load order that works is torch first (libem2lsh.so then binds to the runtime torch mapped,
    same SONAME).  So when torch is installed, import it before mapping libem2lsh.so.  torch stays plumbing:
    nothing of it is called here."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:
        pass


def load():
    """Load libem2lsh.so.  Raises RuntimeError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIBRARY_PATH):
        raise RuntimeError(
            "%s is missing: run `make -C %s` (or __graft_entry__.build()). "
            "There is no fallback implementation." % (LIBRARY_PATH, CSRC_DIR))
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError here == the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    message = load().em2_last_error()
    return message.decode("utf-8", "replace") if message else ""


def check(rc):
    if rc != EM2_OK:
        raise RuntimeError(last_error() or ("em2 error %d" % rc))


def word_count(lsh_count):
    return (int(lsh_count) - 1) // 64 + 1


def _ptr(array):
    return array.ctypes.data_as(ctypes.c_void_p)


def device_count():
    n = ctypes.c_int(0)
    check(load().em2_device_count(ctypes.byref(n)))
    return n.value


def lsh_generate_vectors(gene_count, lsh_count, seed):
    """Lsh::generateLshVectors (src/Lsh.cpp:68-113) -> float64 [gene_count, lsh_count]."""
    out = np.empty((gene_count, lsh_count), dtype=np.float64)
    check(load().em2_lsh_generate_vectors(gene_count, lsh_count, seed, _ptr(out)))
    return out


def similarity_table(lsh_count):
    out = np.empty(lsh_count + 1, dtype=np.float64)
    check(load().em2_lsh_similarity_table(lsh_count, _ptr(out)))
    return out


def murmur_hash_64a(data, seed=231):
    buf = np.ascontiguousarray(data).view(np.uint8)
    return int(load().em2_murmur_hash_64a(_ptr(buf), buf.size, seed))


def make_counts(genes, counts):
    data = np.empty(len(genes), dtype=COUNT_DTYPE)
    data["gene"] = genes
    data["count"] = counts
    return data


def compute_signatures(toc, data, gene_count, vectors, lsh_count):
    """Host-buffer Lsh::computeCellLshSignatures (src/Lsh.cpp:118-224) on the GPU -> uint64 [cells, words]."""
    toc = np.ascontiguousarray(toc, dtype=np.uint64)
    data = np.ascontiguousarray(data, dtype=COUNT_DTYPE)
    vectors = np.ascontiguousarray(vectors, dtype=np.float64)
    cell_count = len(toc) - 1
    assert vectors.shape == (gene_count, lsh_count)
    out = np.zeros((cell_count, word_count(lsh_count)), dtype=np.uint64)
    check(load().em2_compute_signatures(_ptr(toc), _ptr(data), cell_count, gene_count, _ptr(vectors), lsh_count,
                                        _ptr(out)))
    return out


def find_similar_pairs4(signatures, lsh_count, k=100, similarity_threshold=0.2):
    """Host-buffer findSimilarPairs4 pair loop (src/ExpressionMatrixLsh.cpp:200-285) on the GPU.
    Returns (pairs[cells, k] of PAIR_DTYPE, used_count[cells])."""
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    cell_count = signatures.shape[0]
    assert signatures.shape[1] == word_count(lsh_count)
    pairs = np.zeros((cell_count, k), dtype=PAIR_DTYPE)
    used = np.zeros(cell_count, dtype=np.uint32)
    check(load().em2_find_similar_pairs4(_ptr(signatures), cell_count, lsh_count, k, similarity_threshold,
                                         _ptr(pairs), _ptr(used)))
    return pairs, used


def find_similar_pairs5(signatures, lsh_count, k, similarity_threshold, lsh_slice_length, bucket_overflow=1000):
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    cell_count = signatures.shape[0]
    pairs = np.zeros((cell_count, k), dtype=PAIR_DTYPE)
    used = np.zeros(cell_count, dtype=np.uint32)
    check(load().em2_find_similar_pairs5(_ptr(signatures), cell_count, lsh_count, k, similarity_threshold,
                                         lsh_slice_length, bucket_overflow, _ptr(pairs), _ptr(used)))
    return pairs, used


def cell_graph_edges(pairs, used_count, similar_pairs_cell_set, graph_cell_set, similarity_threshold,
                     max_connectivity):
    """CellGraph::CellGraph (src/CellGraph.cpp:33-117) -> (vertex0, vertex1, similarity) per edge, in the order
    the reference adds them; vertex v is graph_cell_set[v]."""
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    used_count = np.ascontiguousarray(used_count, dtype=np.uint32)
    sp_cells = np.ascontiguousarray(similar_pairs_cell_set, dtype=np.uint32)
    graph_cells = np.ascontiguousarray(graph_cell_set, dtype=np.uint32)
    cell_count = len(used_count)
    k = pairs.shape[1] if pairs.ndim == 2 else 0
    if len(sp_cells) != cell_count or (pairs.ndim == 2 and pairs.shape[0] != cell_count):
        raise ValueError("pairs, used_count and the SimilarPairs cell set must describe the same cells")
    per_vertex = k if (max_connectivity == 0 or max_connectivity > k) else int(max_connectivity)
    capacity = max(1, len(graph_cells) * per_vertex)
    v0 = np.zeros(capacity, dtype=np.uint32)
    v1 = np.zeros(capacity, dtype=np.uint32)
    sim = np.zeros(capacity, dtype=np.float32)
    count = ctypes.c_uint64(0)
    check(load().em2_cell_graph_edges(_ptr(pairs), _ptr(used_count), cell_count, k, _ptr(sp_cells), _ptr(graph_cells),
                                      len(graph_cells), similarity_threshold, min(int(max_connectivity), 0xffffffff),
                                      _ptr(v0), _ptr(v1), _ptr(sim), ctypes.byref(count)))
    n = int(count.value)
    return v0[:n].copy(), v1[:n].copy(), sim[:n].copy()


def dev_cell_graph_edges(pairs_ptr, used_ptr, cell_count, k, similar_pairs_cell_set, graph_cell_set, similarity_threshold,
                         max_connectivity):
    """cell_graph_edges with the SimilarPairs content still on the device (device pointers as
    dev_find_similar_pairs4 left them); cell sets in, edges out as host arrays."""
    sp_cells = np.ascontiguousarray(similar_pairs_cell_set, dtype=np.uint32)
    graph_cells = np.ascontiguousarray(graph_cell_set, dtype=np.uint32)
    per_vertex = k if (max_connectivity == 0 or max_connectivity > k) else int(max_connectivity)
    capacity = max(1, len(graph_cells) * per_vertex)
    v0 = np.zeros(capacity, dtype=np.uint32)
    v1 = np.zeros(capacity, dtype=np.uint32)
    sim = np.zeros(capacity, dtype=np.float32)
    count = ctypes.c_uint64(0)
    check(load().em2_dev_cell_graph_edges(pairs_ptr, used_ptr, cell_count, k, _ptr(sp_cells), _ptr(graph_cells),
                                          len(graph_cells), similarity_threshold, min(int(max_connectivity), 0xffffffff),
                                          _ptr(v0), _ptr(v1), _ptr(sim), ctypes.byref(count)))
    n = int(count.value)
    return v0[:n].copy(), v1[:n].copy(), sim[:n].copy()


def dev_cell_graph_edges_to_device(pairs_ptr, used_ptr, cell_count, k, similar_pairs_cell_set, graph_cell_set, similarity_threshold,
                                   max_connectivity, v0_ptr, v1_ptr, sim_ptr):
    """dev_cell_graph_edges with the three edge arrays in DEVICE memory (pointers; room for
    len(graph_cell_set) * min(max_connectivity or k, k) edges each); returns the number of edges."""
    sp_cells = np.ascontiguousarray(similar_pairs_cell_set, dtype=np.uint32)
    graph_cells = np.ascontiguousarray(graph_cell_set, dtype=np.uint32)
    count = ctypes.c_uint64(0)
    check(load().em2_dev_cell_graph_edges(pairs_ptr, used_ptr, cell_count, k, _ptr(sp_cells), _ptr(graph_cells),
                                          len(graph_cells), similarity_threshold, min(int(max_connectivity), 0xffffffff),
                                          v0_ptr, v1_ptr, sim_ptr, ctypes.byref(count)))
    return int(count.value)


def dev_cell_graph_label_propagation(vertex_cell_ids, v0_ptr, v1_ptr, sim_ptr, edge_count, seed=231,
                                     stable_iteration_count_threshold=3, max_iteration_count=100):
    """cell_graph_label_propagation over edge arrays in device memory (as dev_cell_graph_edges_to_device left them)."""
    cells = np.ascontiguousarray(vertex_cell_ids, dtype=np.uint32)
    clusters = np.zeros(len(cells), dtype=np.uint32)
    iterations = ctypes.c_uint64(0)
    check(load().em2_dev_cell_graph_label_propagation(_ptr(cells), len(cells), v0_ptr, v1_ptr, sim_ptr, edge_count, seed,
                                                      stable_iteration_count_threshold, max_iteration_count, _ptr(clusters),
                                                      ctypes.byref(iterations)))
    return clusters, int(iterations.value)


def cell_graph_label_propagation(vertex_cell_ids, edge_vertex0, edge_vertex1, edge_similarity, seed=231,
                                 stable_iteration_count_threshold=3, max_iteration_count=100):
    """CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612) -> (clusterId per vertex, iterations run).
    Host code by nature (a serial schedule defines the result); defaults are ClusterGraphCreationParameters'
    (src/ClusterGraph.hpp:48-50)."""
    cells = np.ascontiguousarray(vertex_cell_ids, dtype=np.uint32)
    v0 = np.ascontiguousarray(edge_vertex0, dtype=np.uint32)
    v1 = np.ascontiguousarray(edge_vertex1, dtype=np.uint32)
    sim = np.ascontiguousarray(edge_similarity, dtype=np.float32)
    if not (len(v0) == len(v1) == len(sim)):
        raise ValueError("the three edge arrays must have one entry per edge")
    clusters = np.zeros(len(cells), dtype=np.uint32)
    iterations = ctypes.c_uint64(0)
    check(load().em2_cell_graph_label_propagation(_ptr(cells), len(cells), _ptr(v0), _ptr(v1), _ptr(sim), len(v0),
                                                  seed, stable_iteration_count_threshold, max_iteration_count,
                                                  _ptr(clusters), ctypes.byref(iterations)))
    return clusters, int(iterations.value)


def analyze_lsh(toc, data, gene_count, signatures, lsh_count, global_cell_ids, seed, csv_downsample, pairs_csv_path,
                statistics_csv_path=None, per_pair=False):
    """ExpressionMatrix::analyzeLsh on a subset's counts and signatures (em2_analyze_lsh) -> dict(sum0, sum1, sum2[, exact,
    lsh]); writes the two csv files."""
    toc = np.ascontiguousarray(toc, dtype=np.uint64)
    data = np.ascontiguousarray(data, dtype=COUNT_DTYPE)
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    ids = np.ascontiguousarray(global_cell_ids, dtype=np.uint32)
    n = len(toc) - 1
    out = {"sum0": np.zeros(200, dtype=np.uint64), "sum1": np.zeros(200, dtype=np.float64), "sum2": np.zeros(200, dtype=np.float64)}
    pairs = n * (n - 1) // 2
    if per_pair:
        out["exact"] = np.zeros(pairs, dtype=np.float64)
        out["lsh"] = np.zeros(pairs, dtype=np.float64)
    check(load().em2_analyze_lsh(_ptr(toc), _ptr(data), n, gene_count, _ptr(signatures), lsh_count, _ptr(ids), seed,
                                 csv_downsample, os.fsencode(pairs_csv_path),
                                 os.fsencode(statistics_csv_path) if statistics_csv_path else None,
                                 _ptr(out["sum0"]), _ptr(out["sum1"]), _ptr(out["sum2"]),
                                 _ptr(out["exact"]) if per_pair else None, _ptr(out["lsh"]) if per_pair else None))
    return out


def find_similar_pairs7(signatures, lsh_count, k, similarity_threshold, lsh_slice_lengths, max_check, log2_bucket_count):
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    lengths = np.ascontiguousarray(lsh_slice_lengths, dtype=np.int32)
    cell_count = signatures.shape[0]
    pairs = np.zeros((cell_count, k), dtype=PAIR_DTYPE)
    used = np.zeros(cell_count, dtype=np.uint32)
    check(load().em2_find_similar_pairs7(_ptr(signatures), cell_count, lsh_count, k, similarity_threshold, _ptr(lengths),
                                         len(lengths), max_check, log2_bucket_count, _ptr(pairs), _ptr(used)))
    return pairs, used


# ---- device-pointer level (torch tensors supply the memory and the stream; this module never imports torch) ----

def dev_find_similar_pairs4_workspace(cell_count, row_count, lsh_count, k):
    return int(load().em2_dev_find_similar_pairs4_workspace(cell_count, row_count, lsh_count, k))


def dev_find_similar_pairs4_form(cell_count, row_count):
    """1 if the scan of this shape runs in its symmetric (each unordered pair once) form, else 0."""
    return int(load().em2_dev_find_similar_pairs4_form(cell_count, row_count))


def dev_find_similar_pairs4_form_for(cell_count, row_count, lsh_count):
    """The form a launch of row_count rows against cell_count columns of lsh_count-bit signatures takes: 0 ordered,
    1 symmetric, 3 symmetric on the matrix cores, 4 rows x all columns on the matrix cores."""
    return int(load().em2_dev_find_similar_pairs4_form_for(cell_count, row_count, lsh_count))


def dev_find_similar_pairs4_last_launch():
    """dict(form, scan_kernel_ms, wave_column_steps, inbox_entries, segments, full_row_cells, matrix_pairs,
    matrix_kernel_ms) of the last launch; form 0 ordered, 1 symmetric, 2 sharded symmetric, 3 symmetric on the matrix cores,
    4 rows x all columns on the matrix cores (a shard of the rows, or the fallback of a symmetric scan)."""
    v = np.zeros(9, dtype=np.float64)
    check(load().em2_dev_find_similar_pairs4_last_launch(_ptr(v), 9))
    return {"form": int(v[0]), "scan_kernel_ms": float(v[1]), "wave_column_steps": float(v[2]),
            "inbox_entries": float(v[3]), "segments": int(v[4]), "full_row_cells": int(v[5]),
            "matrix_pairs": float(v[6]), "matrix_kernel_ms": float(v[7]), "matrix_clock_ghz": float(v[8])}


# ---- em2_collectives (include/em2_lsh.h): the transport table of em2_dist_find_similar_pairs4_with ----
ALL_GATHER_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p)
ALL_REDUCE_MAX_I32_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p)
ALL_TO_ALL_V_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64), _c.c_void_p,
                               _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64), _c.c_void_p)


class Collectives(_c.Structure):
    _fields_ = [("context", _c.c_void_p), ("world", _c.c_int), ("rank", _c.c_int), ("all_gather", ALL_GATHER_FN),
                ("all_reduce_max_i32", ALL_REDUCE_MAX_I32_FN), ("all_to_all_v", ALL_TO_ALL_V_FN)]


DIST_STAGES = ("gather_signatures", "scan", "all_reduce", "exchange", "redistribute")


def dist_find_similar_pairs4_workspace(cell_count, lsh_count, k, rank, world):
    return int(load().em2_dist_find_similar_pairs4_workspace(cell_count, lsh_count, k, rank, world))


def dist_find_similar_pairs4_form(cell_count, lsh_count, k, world):
    return int(load().em2_dist_find_similar_pairs4_form(cell_count, lsh_count, k, world))


def dist_find_similar_pairs4(comm_or_table, local_sig_ptr, cell_count, lsh_count, k, similarity_threshold, all_sig_ptr, pairs_ptr,
                             used_ptr, workspace_ptr, workspace_bytes, stream, timed=False):
    """em2_dist_find_similar_pairs4 (comm_or_table: an ncclComm_t as int) or ..._with (a Collectives table); returns the
    per-stage wall ms as a dict when timed."""
    ms = (_c.c_double * len(DIST_STAGES))() if timed else None
    if isinstance(comm_or_table, Collectives):
        rc = load().em2_dist_find_similar_pairs4_with(_c.addressof(comm_or_table), local_sig_ptr, cell_count, lsh_count, k,
                                                      similarity_threshold, all_sig_ptr, pairs_ptr, used_ptr, workspace_ptr,
                                                      workspace_bytes, stream, ms)
    else:
        rc = load().em2_dist_find_similar_pairs4(comm_or_table, local_sig_ptr, cell_count, lsh_count, k, similarity_threshold,
                                                 all_sig_ptr, pairs_ptr, used_ptr, workspace_ptr, workspace_bytes, stream, ms)
    check(rc)
    return dict(zip(DIST_STAGES, ms)) if timed else None


def dev_find_similar_pairs5_last_launch():
    """dict(gathered_candidates, cells, slice_count, batches, filter_ms, select_ms) of the last findSimilarPairs5 launch."""
    v = np.zeros(7, dtype=np.float64)
    check(load().em2_dev_find_similar_pairs5_last_launch(_ptr(v), 7))
    return {"gathered_candidates": float(v[0]), "cells": int(v[1]), "slice_count": int(v[2]), "batches": int(v[3]),
            "filter_ms": float(v[4]), "select_ms": float(v[5]), "distinct_candidates": float(v[6])}


def dev_release_scratch():
    """Frees the device scratch findSimilarPairs5 keeps between the calls of this process (include/em2_lsh.h)."""
    load().em2_dev_release_scratch()


def dev_find_similar_pairs4(sig_ptr, cell_count, row_begin, row_end, lsh_count, k, similarity_threshold,
                            pairs_ptr, used_ptr, workspace_ptr, workspace_bytes, stream):
    check(load().em2_dev_find_similar_pairs4(sig_ptr, cell_count, row_begin, row_end, lsh_count, k,
                                             similarity_threshold, pairs_ptr, used_ptr, workspace_ptr,
                                             workspace_bytes, stream))


def dev_fsp4_sharded_plan(cell_count, lsh_count, k, rank, world):
    """Layout of one rank's workspace for the sharded symmetric scan (see include/em2_lsh.h)."""
    v = np.zeros(12, dtype=np.uint64)
    check(load().em2_dev_fsp4_sharded_plan(cell_count, lsh_count, k, rank, world, _ptr(v), 12))
    names = ("eligible", "workspace_bytes", "snap_offset", "pool_offset", "pool_capacity", "gathered_offset",
             "gathered_capacity", "prefix_cells", "own_blocks", "blocks", "sorted_offset", "owner_shift")
    return {name: int(x) for name, x in zip(names, v)}


def dev_fsp4_sharded_phase(phase, sig_ptr, cell_count, lsh_count, k, similarity_threshold, rank, world, pairs_ptr,
                           used_ptr, workspace_ptr, workspace_bytes, gathered_count, stream):
    check(load().em2_dev_fsp4_sharded_phase(phase, sig_ptr, cell_count, lsh_count, k, similarity_threshold, rank, world,
                                            pairs_ptr, used_ptr, workspace_ptr, workspace_bytes, gathered_count, stream))


def dev_fsp4_sharded_status(cell_count, k, rank, world, workspace_ptr, stream):
    """(entries in this rank's pool, overflow flag); synchronises the stream."""
    used = ctypes.c_uint64(0)
    overflow = ctypes.c_uint32(0)
    check(load().em2_dev_fsp4_sharded_status(cell_count, k, rank, world, workspace_ptr, stream, ctypes.byref(used),
                                             ctypes.byref(overflow)))
    return int(used.value), int(overflow.value)


def dev_find_similar_pairs4_status(workspace_ptr, row_count, k, stream):
    """Synchronises the stream; raises if the scan reported an incomplete hand-off."""
    check(load().em2_dev_find_similar_pairs4_status(workspace_ptr, row_count, k, stream))


def dev_find_similar_pairs5(sig_ptr, cell_count, row_begin, row_end, lsh_count, k, similarity_threshold,
                            lsh_slice_length, bucket_overflow, pairs_ptr, used_ptr, stream):
    check(load().em2_dev_find_similar_pairs5(sig_ptr, cell_count, row_begin, row_end, lsh_count, k,
                                             similarity_threshold, lsh_slice_length, bucket_overflow, pairs_ptr,
                                             used_ptr, stream))


def dev_compute_signatures_workspace(cell_count, lsh_count):
    return int(load().em2_dev_compute_signatures_workspace(cell_count, lsh_count))


def dev_compute_signatures(toc_ptr, data_ptr, cell_count, gene_count, vectors_ptr, vector_aux_ptr, lsh_count,
                           sig_ptr, workspace_ptr, workspace_bytes, stream):
    check(load().em2_dev_compute_signatures(toc_ptr, data_ptr, cell_count, gene_count, vectors_ptr,
                                            vector_aux_ptr, lsh_count, sig_ptr, workspace_ptr, workspace_bytes,
                                            stream))


TIER_NAMES = ("exact", "float", "fixed16-float", "fixed16-integer")


def dev_compute_signatures_tier(workspace_ptr, cell_count, lsh_count, have_vector_aux=True):
    """Which first tier the last dev_compute_signatures call on this workspace ran (include/em2_lsh.h: EM2_TIER_*), by name."""
    tier = ctypes.c_int(-1)
    check(load().em2_dev_compute_signatures_tier(workspace_ptr, cell_count, lsh_count, 1 if have_vector_aux else 0, ctypes.byref(tier)))
    return TIER_NAMES[tier.value]


def dev_vector_aux_bytes(gene_count, lsh_count):
    return int(load().em2_dev_vector_aux_bytes(gene_count, lsh_count))


def dev_prepare_vectors(vectors_ptr, gene_count, lsh_count, aux_ptr, stream):
    check(load().em2_dev_prepare_vectors(vectors_ptr, gene_count, lsh_count, aux_ptr, stream))

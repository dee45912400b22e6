"""The reference's on-disk objects of the LSH path, through the C ABI: SimilarPairs-<name>-{Info,Pairs,CellInfo}
(src/SimilarPairs.cpp), Lsh-<name>-{Info,Signatures} (src/Lsh.cpp), plus test/bench tooling that creates a data
directory holding exactly the files the path reads."""
import ctypes

import numpy as np

from . import capi


def _b(s):
    return s.encode("utf-8") if isinstance(s, str) else s


def write_similar_pairs(directory, name, gene_set_name, cell_set_name, k, pairs, used_count):
    pairs = np.ascontiguousarray(pairs, dtype=capi.PAIR_DTYPE)
    used_count = np.ascontiguousarray(used_count, dtype=np.uint32)
    capi.check(capi.load().em2_similar_pairs_write(_b(directory), _b(name), _b(gene_set_name), _b(cell_set_name), k,
                                                   len(used_count), capi._ptr(pairs), capi._ptr(used_count)))


def read_similar_pairs(directory, name):
    """SimilarPairs(directory, name) of src/SimilarPairs.cpp:47-83 -> (k, pairs[cells, k], used_count[cells])."""
    lib = capi.load()
    k = ctypes.c_uint64(0)
    cells = ctypes.c_uint64(0)
    capi.check(lib.em2_similar_pairs_read(_b(directory), _b(name), ctypes.byref(k), ctypes.byref(cells), None, None))
    pairs = np.zeros((cells.value, k.value), dtype=capi.PAIR_DTYPE)
    used = np.zeros(cells.value, dtype=np.uint32)
    capi.check(lib.em2_similar_pairs_read(_b(directory), _b(name), ctypes.byref(k), ctypes.byref(cells),
                                          capi._ptr(pairs), capi._ptr(used)))
    return int(k.value), pairs, used


def similar_pairs_info(directory, name):
    """SimilarPairs::Info (src/SimilarPairs.hpp:188-198) -> (k, cellCount, geneSetName, cellSetName)."""
    k = ctypes.c_uint64(0)
    cells = ctypes.c_uint64(0)
    gene_set = ctypes.create_string_buffer(256)
    cell_set = ctypes.create_string_buffer(256)
    capi.check(capi.load().em2_similar_pairs_info(_b(directory), _b(name), ctypes.byref(k), ctypes.byref(cells),
                                                  gene_set, cell_set))
    return int(k.value), int(cells.value), gene_set.value.decode("utf-8"), cell_set.value.decode("utf-8")


def write_lsh(directory, lsh_name, lsh_count, signatures):
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    capi.check(capi.load().em2_lsh_write(_b(directory), _b(lsh_name), signatures.shape[0], lsh_count,
                                         capi._ptr(signatures)))


def read_lsh(directory, lsh_name):
    """Lsh(name) of src/Lsh.cpp:48-64 -> (lsh_count, signatures[cells, words])."""
    lib = capi.load()
    cells = ctypes.c_uint64(0)
    lsh_count = ctypes.c_uint64(0)
    capi.check(lib.em2_lsh_read(_b(directory), _b(lsh_name), ctypes.byref(cells), ctypes.byref(lsh_count), None))
    sig = np.zeros((cells.value, capi.word_count(lsh_count.value)), dtype=np.uint64)
    capi.check(lib.em2_lsh_read(_b(directory), _b(lsh_name), ctypes.byref(cells), ctypes.byref(lsh_count),
                                capi._ptr(sig)))
    return int(lsh_count.value), sig


# ---- tooling (not a reference API) ----

def create_directory(directory, gene_count, toc, data):
    toc = np.ascontiguousarray(toc, dtype=np.uint64)
    data = np.ascontiguousarray(data, dtype=capi.COUNT_DTYPE)
    capi.check(capi.load().em2_tool_create_directory(_b(directory), gene_count, len(toc) - 1, capi._ptr(toc),
                                                     capi._ptr(data)))


def add_gene_set(directory, name, sorted_global_ids):
    ids = np.ascontiguousarray(sorted_global_ids, dtype=np.uint32)
    capi.check(capi.load().em2_tool_add_gene_set(_b(directory), _b(name), capi._ptr(ids), len(ids)))


def add_cell_set(directory, name, sorted_cell_ids):
    ids = np.ascontiguousarray(sorted_cell_ids, dtype=np.uint32)
    capi.check(capi.load().em2_tool_add_cell_set(_b(directory), _b(name), capi._ptr(ids), len(ids)))

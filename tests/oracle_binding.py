"""ctypes bindings of the CHECKERS used by the tests: the CPU oracle (oracle/libem2oracle.so), the reference's
own Boost-free components (oracle/_ref/libem2ref.so, only where /root/reference exists) and a host build of
the product's shared headers (tests/native).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NATIVE_DIR = os.path.join(ROOT, "tests", "native")

c = ctypes
P = c.c_void_p


def _ptr(a):
    return a.ctypes.data_as(c.c_void_p)


def _make(target=None):
    cmd = ["make", "-C", ORACLE_DIR]
    if target:
        cmd.append(target)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed: " + r.stderr)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.em2o_generate_lsh_vectors.argtypes = [c.c_uint32, c.c_uint32, c.c_uint32, P]
        lib.em2o_similarity_table.argtypes = [c.c_uint32, P]
        lib.em2o_compute_signatures.argtypes = [P, P, P, c.c_uint32, c.c_uint32, P, c.c_uint32, P]
        lib.em2o_mismatch_matrix.argtypes = [P, c.c_uint32, c.c_uint32, P]
        lib.em2o_find_similar_pairs4.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double, P, P, P]
        lib.em2o_find_similar_pairs4_rows.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double,
                                                      c.c_uint32, c.c_uint32, P, P, P]
        lib.em2o_find_similar_pairs5.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double, c.c_uint32,
                                                 c.c_uint64, P, P, P]
        lib.em2o_find_similar_pairs5.restype = c.c_int
        lib.em2o_find_similar_pairs5_rows.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double, c.c_uint32,
                                                      c.c_uint64, c.c_uint32, c.c_uint32, P, P, P]
        lib.em2o_find_similar_pairs5_rows.restype = c.c_int
        lib.em2o_find_similar_pairs5_cells.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double, c.c_uint32,
                                                       c.c_uint64, P, c.c_uint32, P, P, P]
        lib.em2o_find_similar_pairs5_cells.restype = c.c_int
        lib.em2o_keep_best.argtypes = [P, P, c.c_uint32, c.c_uint32]
        lib.em2o_keep_best.restype = c.c_uint32
        lib.em2o_multiple_set_union.argtypes = [P, P, c.c_uint32, P]
        lib.em2o_multiple_set_union.restype = c.c_uint32
        lib.em2o_find_similar_pairs7.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double, P, c.c_uint32,
                                                 c.c_uint32, c.c_uint32, P, P, P]
        lib.em2o_find_similar_pairs7.restype = c.c_int
        lib.em2o_cell_graph_edges.argtypes = [P, P, c.c_uint32, c.c_uint32, P, P, c.c_uint32, c.c_double, c.c_uint64,
                                              P, P, P]
        lib.em2o_cell_graph_edges.restype = c.c_uint64
        lib.em2o_cell_graph_edges_hashed.argtypes = lib.em2o_cell_graph_edges.argtypes
        lib.em2o_cell_graph_edges_hashed.restype = c.c_uint64
        lib.em2o_label_propagation.argtypes = [P, c.c_uint32, P, P, P, c.c_uint64, c.c_uint64, c.c_uint64, c.c_uint64, P]
        lib.em2o_label_propagation.restype = c.c_uint64
        lib.em2o_murmur_hash_64a.argtypes = [P, c.c_int, c.c_uint64]
        lib.em2o_murmur_hash_64a.restype = c.c_uint64
        lib.em2o_subset.argtypes = [P, P, P, P, c.c_uint32, P, c.c_uint32, P, c.c_uint32, P, P, P, P]
        lib.em2o_subset.restype = c.c_int64

    def subset(self, toc, genes, counts, gene_set, local_gene_ids, cell_set):
        """ExpressionMatrixSubset ctor + computeSums -> (toc, genes, counts, sums[cells, 2]); ValueError where the
        reference's CZI_ASSERT(is_sorted) throws."""
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        genes = np.ascontiguousarray(genes, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.float32)
        gene_set = np.ascontiguousarray(gene_set, dtype=np.uint32)
        local_gene_ids = np.ascontiguousarray(local_gene_ids, dtype=np.uint32)
        cell_set = np.ascontiguousarray(cell_set, dtype=np.uint32)
        args = (_ptr(toc), _ptr(genes), _ptr(counts), _ptr(gene_set), len(gene_set), _ptr(local_gene_ids),
                len(local_gene_ids), _ptr(cell_set), len(cell_set))
        n = self.lib.em2o_subset(*args, None, None, None, None)
        if n < 0:
            raise ValueError("gene set or cell set not sorted (ExpressionMatrixSubset.cpp:17-18)")
        out_toc = np.zeros(len(cell_set) + 1, dtype=np.uint64)
        out_genes = np.zeros(n, dtype=np.uint32)
        out_counts = np.zeros(n, dtype=np.float32)
        sums = np.zeros((len(cell_set), 2), dtype=np.float64)
        n2 = self.lib.em2o_subset(*args, _ptr(out_toc), _ptr(out_genes), _ptr(out_counts), _ptr(sums))
        assert n2 == n
        return out_toc, out_genes, out_counts, sums

    def analyze_lsh(self, toc, genes, counts, gene_count, sig, lsh_count, global_cell_ids, seed, csv_downsample,
                    pairs_csv_path, statistics_csv_path):
        """ExpressionMatrix::analyzeLsh -> dict(sum0, sum1, sum2, exact, lsh), or None where CZI_ASSERT(bin < binCount) throws."""
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        genes = np.ascontiguousarray(genes, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.float32)
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        ids = np.ascontiguousarray(global_cell_ids, dtype=np.uint32)
        n = len(toc) - 1
        sums = np.zeros(2 * n, dtype=np.float64)
        for c in range(n):                       # ExpressionMatrixSubset::computeSums (:47-58): float values, sequential double sums
            v = counts[int(toc[c]):int(toc[c + 1])]
            if len(v):
                sums[2 * c] = np.cumsum(v.astype(np.float64))[-1]                 # cumsum adds left to right
                sums[2 * c + 1] = np.cumsum((v * v).astype(np.float64))[-1]       # float product, then double
        pairs = n * (n - 1) // 2
        out = {"sum0": np.zeros(200, dtype=np.uint64), "sum1": np.zeros(200), "sum2": np.zeros(200),
               "exact": np.zeros(max(1, pairs)), "lsh": np.zeros(max(1, pairs))}
        self.lib.em2o_analyze_lsh.restype = ctypes.c_int64
        self.lib.em2o_analyze_lsh.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                              ctypes.c_void_p, ctypes.c_uint32, ctypes.c_double, ctypes.c_char_p, ctypes.c_char_p] + \
                                             [ctypes.c_void_p] * 5
        done = self.lib.em2o_analyze_lsh(_ptr(toc), _ptr(genes), _ptr(counts), _ptr(sums), n, gene_count, _ptr(sig), lsh_count, _ptr(ids),
                                         seed, csv_downsample, os.fsencode(pairs_csv_path), os.fsencode(statistics_csv_path),
                                         _ptr(out["sum0"]), _ptr(out["sum1"]), _ptr(out["sum2"]), _ptr(out["exact"]), _ptr(out["lsh"]))
        if done < 0:
            return None
        assert done == pairs
        out["exact"], out["lsh"] = out["exact"][:pairs], out["lsh"][:pairs]
        return out

    def generate_lsh_vectors(self, gene_count, lsh_count, seed):
        out = np.empty((gene_count, lsh_count), dtype=np.float64)
        self.lib.em2o_generate_lsh_vectors(gene_count, lsh_count, seed, _ptr(out))
        return out

    def similarity_table(self, lsh_count):
        out = np.empty(lsh_count + 1, dtype=np.float64)
        self.lib.em2o_similarity_table(lsh_count, _ptr(out))
        return out

    def compute_signatures(self, toc, genes, counts, gene_count, vectors, lsh_count):
        toc = np.ascontiguousarray(toc, dtype=np.uint64)
        genes = np.ascontiguousarray(genes, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.float32)
        vectors = np.ascontiguousarray(vectors, dtype=np.float64)
        n = len(toc) - 1
        out = np.zeros((n, (lsh_count - 1) // 64 + 1), dtype=np.uint64)
        self.lib.em2o_compute_signatures(_ptr(toc), _ptr(genes), _ptr(counts), n, gene_count, _ptr(vectors),
                                         lsh_count, _ptr(out))
        return out

    def mismatch_matrix(self, sig, lsh_count):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n = sig.shape[0]
        out = np.zeros((n, n), dtype=np.uint16)
        self.lib.em2o_mismatch_matrix(_ptr(sig), n, lsh_count, _ptr(out))
        return out

    def find_similar_pairs4(self, sig, lsh_count, k, thr):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n = sig.shape[0]
        cell = np.zeros((n, k), dtype=np.uint32)
        sim = np.zeros((n, k), dtype=np.float32)
        used = np.zeros(n, dtype=np.uint32)
        self.lib.em2o_find_similar_pairs4(_ptr(sig), n, lsh_count, k, thr, _ptr(cell), _ptr(sim), _ptr(used))
        return cell, sim, used

    def find_similar_pairs4_rows(self, sig, lsh_count, k, thr, row_begin, row_end):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n = sig.shape[0]
        rows = row_end - row_begin
        cell = np.zeros((rows, k), dtype=np.uint32)
        sim = np.zeros((rows, k), dtype=np.float32)
        used = np.zeros(rows, dtype=np.uint32)
        self.lib.em2o_find_similar_pairs4_rows(_ptr(sig), n, lsh_count, k, thr, row_begin, row_end, _ptr(cell),
                                               _ptr(sim), _ptr(used))
        return cell, sim, used

    def find_similar_pairs5(self, sig, lsh_count, k, thr, slice_length, bucket_overflow):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n = sig.shape[0]
        cell = np.zeros((n, k), dtype=np.uint32)
        sim = np.zeros((n, k), dtype=np.float32)
        used = np.zeros(n, dtype=np.uint32)
        rc = self.lib.em2o_find_similar_pairs5(_ptr(sig), n, lsh_count, k, thr, slice_length, bucket_overflow,
                                               _ptr(cell), _ptr(sim), _ptr(used))
        if rc != 0:
            raise ValueError("oracle fsp5 rejected the arguments")
        return cell, sim, used

    def find_similar_pairs5_rows(self, sig, lsh_count, k, thr, slice_length, bucket_overflow, row_begin, row_end):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        rows = row_end - row_begin
        cell = np.zeros((rows, k), dtype=np.uint32)
        sim = np.zeros((rows, k), dtype=np.float32)
        used = np.zeros(rows, dtype=np.uint32)
        rc = self.lib.em2o_find_similar_pairs5_rows(_ptr(sig), sig.shape[0], lsh_count, k, thr, slice_length,
                                                    bucket_overflow, row_begin, row_end, _ptr(cell), _ptr(sim),
                                                    _ptr(used))
        if rc != 0:
            raise ValueError("oracle fsp5 rejected the arguments")
        return cell, sim, used

    def find_similar_pairs5_cells(self, sig, lsh_count, k, thr, slice_length, bucket_overflow, cells):
        """findSimilarPairs5 for the listed cells only (the tables, over all cells, are built once)."""
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
        cell = np.zeros((len(cells), k), dtype=np.uint32)
        sim = np.zeros((len(cells), k), dtype=np.float32)
        used = np.zeros(len(cells), dtype=np.uint32)
        rc = self.lib.em2o_find_similar_pairs5_cells(_ptr(sig), sig.shape[0], lsh_count, k, thr, slice_length,
                                                     bucket_overflow, _ptr(cells), len(cells), _ptr(cell), _ptr(sim), _ptr(used))
        if rc != 0:
            raise ValueError("oracle fsp5 rejected the arguments")
        return cell, sim, used

    def find_similar_pairs7(self, sig, lsh_count, k, thr, slice_lengths, max_check, log2_bucket_count):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n = sig.shape[0]
        lengths = np.ascontiguousarray(slice_lengths, dtype=np.int32)
        cell = np.zeros((n, k), dtype=np.uint32)
        sim = np.zeros((n, k), dtype=np.float32)
        used = np.zeros(n, dtype=np.uint32)
        rc = self.lib.em2o_find_similar_pairs7(_ptr(sig), n, lsh_count, k, thr, _ptr(lengths), len(lengths), max_check,
                                               log2_bucket_count, _ptr(cell), _ptr(sim), _ptr(used))
        if rc != 0:
            raise ValueError("oracle fsp7 rejected the arguments (%d)" % rc)
        return cell, sim, used

    def cell_graph_edges(self, cell, sim, used, sp_cells, graph_cells, thr, max_connectivity, hashed=False):
        """hashed=True: the same loop with hash tables in place of the std::map / std::set (million-cell problems)."""
        n, k = cell.shape
        pairs = np.zeros((n, k), dtype=np.dtype([("cell", np.uint32), ("similarity", np.float32)]))
        pairs["cell"] = cell
        pairs["similarity"] = sim
        used = np.ascontiguousarray(used, dtype=np.uint32)
        sp_cells = np.ascontiguousarray(sp_cells, dtype=np.uint32)
        graph_cells = np.ascontiguousarray(graph_cells, dtype=np.uint32)
        cap = max(1, len(graph_cells) * k)
        v0 = np.zeros(cap, dtype=np.uint32)
        v1 = np.zeros(cap, dtype=np.uint32)
        es = np.zeros(cap, dtype=np.float32)
        function = self.lib.em2o_cell_graph_edges_hashed if hashed else self.lib.em2o_cell_graph_edges
        m = function(_ptr(pairs), _ptr(used), n, k, _ptr(sp_cells), _ptr(graph_cells),
                     len(graph_cells), thr, max_connectivity, _ptr(v0), _ptr(v1), _ptr(es))
        return v0[:m], v1[:m], es[:m]

    def label_propagation(self, vertex_cells, v0, v1, sim, seed=231, stable=3, max_iterations=100):
        vertex_cells = np.ascontiguousarray(vertex_cells, dtype=np.uint32)
        v0 = np.ascontiguousarray(v0, dtype=np.uint32)
        v1 = np.ascontiguousarray(v1, dtype=np.uint32)
        sim = np.ascontiguousarray(sim, dtype=np.float32)
        out = np.zeros(len(vertex_cells), dtype=np.uint32)
        iterations = self.lib.em2o_label_propagation(_ptr(vertex_cells), len(vertex_cells), _ptr(v0), _ptr(v1), _ptr(sim),
                                                     len(v0), seed, stable, max_iterations, _ptr(out))
        return out, int(iterations)

    def keep_best(self, cell, sim, k):
        cell = np.array(cell, dtype=np.uint32)
        sim = np.array(sim, dtype=np.float32)
        n = self.lib.em2o_keep_best(_ptr(cell), _ptr(sim), len(cell), k)
        return cell[:n], sim[:n]

    def multiple_set_union(self, sets):
        values = np.concatenate([np.asarray(s, dtype=np.uint32) for s in sets]) if sets else np.zeros(0, np.uint32)
        offsets = np.zeros(len(sets) + 1, dtype=np.uint32)
        offsets[1:] = np.cumsum([len(s) for s in sets])
        out = np.zeros(max(1, len(values)), dtype=np.uint32)
        n = self.lib.em2o_multiple_set_union(_ptr(values), _ptr(offsets), len(sets), _ptr(out))
        return out[:n]

    def murmur(self, data, seed=231):
        buf = np.ascontiguousarray(data).view(np.uint8)
        return int(self.lib.em2o_murmur_hash_64a(_ptr(buf), buf.size, seed))


class Ref:
    def __init__(self, lib):
        self.lib = lib
        lib.em2ref_keep_best.argtypes = [P, P, c.c_uint32, c.c_uint32]
        lib.em2ref_keep_best.restype = c.c_uint32
        lib.em2ref_keep_best_int_greater.argtypes = [P, c.c_uint32, c.c_uint32]
        lib.em2ref_keep_best_int_greater.restype = c.c_uint32
        lib.em2ref_sort_pairs.argtypes = [P, P, c.c_uint32]
        lib.em2ref_murmur_hash_64a.argtypes = [P, c.c_int, c.c_uint64]
        lib.em2ref_murmur_hash_64a.restype = c.c_uint64

    def keep_best(self, cell, sim, k):
        cell = np.array(cell, dtype=np.uint32)
        sim = np.array(sim, dtype=np.float32)
        n = self.lib.em2ref_keep_best(_ptr(cell), _ptr(sim), len(cell), k)
        return cell[:n], sim[:n]

    def keep_best_int_greater(self, values, k):
        v = np.array(values, dtype=np.int32)
        n = self.lib.em2ref_keep_best_int_greater(_ptr(v), len(v), k)
        return v[:n]

    def sort_pairs(self, cell, sim):
        cell = np.array(cell, dtype=np.uint32)
        sim = np.array(sim, dtype=np.float32)
        self.lib.em2ref_sort_pairs(_ptr(cell), _ptr(sim), len(cell))
        return cell, sim

    def murmur(self, data, seed=231):
        buf = np.ascontiguousarray(data).view(np.uint8)
        return int(self.lib.em2ref_murmur_hash_64a(_ptr(buf), buf.size, seed))


class HostChecks:
    def __init__(self, lib):
        self.lib = lib
        lib.em2t_nth_element.argtypes = [P, P, c.c_uint32, c.c_uint32, c.c_int]
        lib.em2t_std_introselect.argtypes = [P, P, c.c_uint32, c.c_uint32, c.c_int]
        lib.em2t_wave_model_nth_element.argtypes = [P, P, c.c_uint32, c.c_uint32, c.c_int]
        lib.em2t_tables.argtypes = [c.c_uint32, c.c_double, P, P, P, P, c.POINTER(c.c_int32), c.POINTER(c.c_int32)]
        lib.em2t_tables.restype = c.c_uint32
        lib.em2t_fsp4_rows.argtypes = [P, c.c_uint32, c.c_uint32, c.c_uint32, c.c_double, c.c_uint32, c.c_uint32,
                                       P, P, P]
        lib.em2t_fsp4_rows.restype = c.c_int

    def nth_element(self, cell, key, nth, depth_limit=-1):
        cell = np.array(cell, dtype=np.uint32)
        key = np.array(key, dtype=np.uint32)
        self.lib.em2t_nth_element(_ptr(cell), _ptr(key), len(cell), nth, depth_limit)
        return cell, key

    def wave_model_nth_element(self, cell, key, nth, depth_limit=-1):
        cell = np.array(cell, dtype=np.uint32)
        key = np.array(key, dtype=np.uint32)
        self.lib.em2t_wave_model_nth_element(_ptr(cell), _ptr(key), len(cell), nth, depth_limit)
        return cell, key

    def std_introselect(self, cell, key, nth, depth_limit=-1):
        cell = np.array(cell, dtype=np.uint32)
        key = np.array(key, dtype=np.uint32)
        self.lib.em2t_std_introselect(_ptr(cell), _ptr(key), len(cell), nth, depth_limit)
        return cell, key

    def tables(self, lsh_count, thr):
        sim = np.zeros(lsh_count + 1, dtype=np.float64)
        key_of_m = np.zeros(lsh_count + 1, dtype=np.uint32)
        key_sim = np.zeros(lsh_count + 1, dtype=np.float32)
        accept = np.zeros(lsh_count + 1, dtype=np.int32)
        mg = c.c_int32(0)
        m0 = c.c_int32(0)
        kc = self.lib.em2t_tables(lsh_count, thr, _ptr(sim), _ptr(key_of_m), _ptr(key_sim), _ptr(accept),
                                  c.byref(mg), c.byref(m0))
        assert kc > 0
        return dict(similarity=sim, key_of_mismatch=key_of_m, key_similarity=key_sim[:kc],
                    accept_max_by_key=accept[:kc], m_global=mg.value, m_max_initial=m0.value)

    def fsp4_rows(self, sig, lsh_count, k, thr, row_begin, row_end):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        rows = row_end - row_begin
        cell = np.zeros((rows, k), dtype=np.uint32)
        sim = np.zeros((rows, k), dtype=np.float32)
        used = np.zeros(rows, dtype=np.uint32)
        rc = self.lib.em2t_fsp4_rows(_ptr(sig), sig.shape[0], lsh_count, k, thr, row_begin, row_end, _ptr(cell),
                                     _ptr(sim), _ptr(used))
        assert rc == 0
        return cell, sim, used


def load_oracle():
    path = os.path.join(ORACLE_DIR, "libem2oracle.so")
    src = os.path.join(ORACLE_DIR, "em2_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        _make()
    return Oracle(ctypes.CDLL(path))


def load_ref():
    path = os.path.join(ORACLE_DIR, "_ref", "libem2ref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            _make("ref")
        else:
            return None
    return Ref(ctypes.CDLL(path))


class RefLayout:
    """oracle/_ref/libem2ref_layout.so: SimilarPairs::Info built from the reference's own StaticString255 (oracle/ref_layout.cpp)."""

    def __init__(self, lib):
        self.lib = lib
        lib.em2ref_info_size.restype = c.c_uint64
        lib.em2ref_info_offsets.argtypes = [P]
        lib.em2ref_make_info.argtypes = [c.c_uint64, c.c_char_p, c.c_uint64, c.c_char_p, c.c_uint64, P]
        lib.em2ref_make_info.restype = c.c_int

    def info_size(self):
        return int(self.lib.em2ref_info_size())

    def info_offsets(self):
        out = np.zeros(8, dtype=np.uint64)
        self.lib.em2ref_info_offsets(_ptr(out))
        return [int(x) for x in out]

    def make_info(self, k, gene_set_name, gene_set_hash, cell_set_name, cell_set_hash):
        out = np.zeros(self.info_size(), dtype=np.uint8)
        rc = self.lib.em2ref_make_info(k, gene_set_name.encode(), gene_set_hash, cell_set_name.encode(), cell_set_hash, _ptr(out))
        if rc != 0:
            raise ValueError("ShortStaticString capacity exceeded.")
        return out.tobytes()


def load_ref_layout():
    path = os.path.join(ORACLE_DIR, "_ref", "libem2ref_layout.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            _make("ref")
        else:
            return None
    return RefLayout(ctypes.CDLL(path))


def load_host_checks():
    build = os.path.join(NATIVE_DIR, "build")
    os.makedirs(build, exist_ok=True)
    path = os.path.join(build, "libem2hostchecks.so")
    sources = [os.path.join(NATIVE_DIR, "em2_host_checks.cpp"),
               os.path.join(ROOT, "expressionmatrix2_amd", "csrc", "em2_tables.cpp")]
    deps = sources + [os.path.join(ROOT, "expressionmatrix2_amd", "csrc", "em2_select.h"),
                      os.path.join(ROOT, "expressionmatrix2_amd", "csrc", "em2_tables.h")]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(d) for d in deps):
        cmd = ["g++", "-std=c++17", "-O2", "-msse4.2", "-ffp-contract=off", "-fPIC", "-shared", "-o", path] + sources
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host checks build failed: " + r.stderr)
    return HostChecks(ctypes.CDLL(path))

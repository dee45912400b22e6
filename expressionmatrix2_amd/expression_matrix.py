"""ExpressionMatrix: the reference's Python-bound class (src/PythonModule.cpp:158-1275) restricted to the methods of
the LSH similar-pairs path, with the same names, keyword arguments, defaults and error behaviour.

    e = ExpressionMatrix(directoryName)
    e.findSimilarPairs4(similarPairsName="Lsh")                       # tests/CaseStudy1/compute1.py:12
    e.computeLshSignatures(lshName="L"); e.findSimilarPairs5(lshName="L", similarPairsName="P", lshSliceLength=16)

Like the reference the methods return None and leave their result as files in the data directory
(SimilarPairs-<name>-*, Lsh-<name>-*), addressed by name; errors surface as RuntimeError with the reference's
message text.  The work happens in libem2lsh.so (HIP, MI355X); when torch.distributed is initialised with more
than one rank the calls are collective and the scan is row-sharded over the ranks (sharded.py)."""
import ctypes

import numpy as np

from . import capi, files

_REQUIRED = object()


def _b(s):
    if not isinstance(s, str):
        raise TypeError("expected a string, got %r" % (s,))
    return s.encode("utf-8")


def _distributed():
    """(dist module, world_size) if torch.distributed is up with more than one rank, else (None, 1)."""
    import sys
    if "torch" not in sys.modules:
        return None, 1
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_world_size()
    return None, 1


class ExpressionMatrix:
    def __init__(self, directoryName, allowReadOnly=False):
        # src/PythonModule.cpp:164-176 / src/ExpressionMatrix.cpp:39-52.  The reference creates the directory when it
        # does not exist; creating an EMPTY expression matrix is ingest (out of scope here), so a missing directory
        # is an error.  allowReadOnly is accepted for signature compatibility: the path never writes its inputs.
        self.directoryName = directoryName
        self._handle = ctypes.c_void_p(None)
        capi.check(capi.load().em2_matrix_open(_b(directoryName), ctypes.byref(self._handle)))

    def close(self):
        if self._handle:
            capi.load().em2_matrix_close(self._handle)
            self._handle = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- src/PythonModule.cpp:802-824 ----
    def findSimilarPairs4(self, geneSetName="AllGenes", cellSetName="AllCells", similarPairsName=_REQUIRED, k=100,
                          similarityThreshold=0.2, lshCount=1024, seed=231):
        if similarPairsName is _REQUIRED:
            raise TypeError("findSimilarPairs4(): missing required argument 'similarPairsName'")
        dist, world = _distributed()
        if dist is not None:
            from . import sharded
            return sharded.find_similar_pairs4_collective(self, geneSetName, cellSetName, similarPairsName, k,
                                                          similarityThreshold, lshCount, seed, dist)
        capi.check(capi.load().em2_matrix_find_similar_pairs4(self._handle, _b(geneSetName), _b(cellSetName),
                                                              _b(similarPairsName), k, similarityThreshold,
                                                              lshCount, seed))

    # ---- src/PythonModule.cpp:945-953 ----
    def computeLshSignatures(self, geneSetName="AllGenes", cellSetName="AllCells", lshName=_REQUIRED, lshCount=1024,
                             seed=231):
        if lshName is _REQUIRED:
            raise TypeError("computeLshSignatures(): missing required argument 'lshName'")
        capi.check(capi.load().em2_matrix_compute_lsh_signatures(self._handle, _b(geneSetName), _b(cellSetName),
                                                                 _b(lshName), lshCount, seed))

    # ---- src/PythonModule.cpp:852-865 ----
    def findSimilarPairs5(self, geneSetName="AllGenes", cellSetName="AllCells", lshName=_REQUIRED,
                          similarPairsName=_REQUIRED, k=100, similarityThreshold=0.2, lshSliceLength=_REQUIRED,
                          bucketOverflow=1000):
        for name, value in (("lshName", lshName), ("similarPairsName", similarPairsName),
                            ("lshSliceLength", lshSliceLength)):
            if value is _REQUIRED:
                raise TypeError("findSimilarPairs5(): missing required argument '%s'" % name)
        dist, world = _distributed()
        if dist is not None:
            from . import sharded
            return sharded.find_similar_pairs5_collective(self, geneSetName, cellSetName, lshName, similarPairsName,
                                                          k, similarityThreshold, lshSliceLength, bucketOverflow,
                                                          dist)
        capi.check(capi.load().em2_matrix_find_similar_pairs5(self._handle, _b(geneSetName), _b(cellSetName),
                                                              _b(lshName), _b(similarPairsName), k,
                                                              similarityThreshold, lshSliceLength, bucketOverflow))

    # ---- src/PythonModule.cpp:926-934 ----
    def removeSimilarPairs(self, similarPairsName):
        capi.check(capi.load().em2_matrix_remove_similar_pairs(self._handle, _b(similarPairsName)))

    def _subset_sizes(self, geneSetName, cellSetName):
        """(geneCount, cellCount, nnz) of the subset; raises the reference's lookup / emptiness errors."""
        genes = ctypes.c_uint32(0)
        cells = ctypes.c_uint32(0)
        nnz = ctypes.c_uint64(0)
        capi.check(capi.load().em2_matrix_subset(self._handle, _b(geneSetName), _b(cellSetName), ctypes.byref(genes),
                                                 ctypes.byref(cells), ctypes.byref(nnz), None, None))
        return int(genes.value), int(cells.value), int(nnz.value)

    # ---- helper used by the sharded driver and by tests (ExpressionMatrixSubset as arrays) ----
    def _subset(self, geneSetName, cellSetName):
        lib = capi.load()
        genes = ctypes.c_uint32(0)
        cells = ctypes.c_uint32(0)
        nnz = ctypes.c_uint64(0)
        capi.check(lib.em2_matrix_subset(self._handle, _b(geneSetName), _b(cellSetName), ctypes.byref(genes),
                                         ctypes.byref(cells), ctypes.byref(nnz), None, None))
        toc = np.zeros(cells.value + 1, dtype=np.uint64)
        data = np.zeros(nnz.value, dtype=capi.COUNT_DTYPE)
        capi.check(lib.em2_matrix_subset(self._handle, _b(geneSetName), _b(cellSetName), ctypes.byref(genes),
                                         ctypes.byref(cells), ctypes.byref(nnz), capi._ptr(toc), capi._ptr(data)))
        return int(genes.value), toc, data

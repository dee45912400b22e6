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
        self._cellGraphs = {}          # ExpressionMatrix::cellGraphs (src/ExpressionMatrix.hpp): in memory only
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

    # ---- src/PythonModule.cpp:940-944: bound without argument names or defaults ("Only intended to be used for testing") ----
    def analyzeLsh(self, geneSetName, cellSetName, lshCount, seed, csvDownsample):
        """Writes Lsh-analysis.csv and LSH-analysis-statistics.csv into the working directory
        (src/ExpressionMatrixLsh.cpp:1303, :1345)."""
        capi.check(capi.load().em2_matrix_analyze_lsh(self._handle, _b(geneSetName), _b(cellSetName), lshCount, seed,
                                                      csvDownsample, None))

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

    # ---- src/PythonModule.cpp:882-897 ("Prototype code. Use findSimilarPairs4 instead.") ----
    def findSimilarPairs7(self, geneSetName="AllGenes", cellSetName="AllCells", lshName=_REQUIRED,
                          similarPairsName=_REQUIRED, k=100, similarityThreshold=0.2, lshSliceLengths=_REQUIRED,
                          maxCheck=_REQUIRED, log2BucketCount=_REQUIRED):
        if _REQUIRED in (lshName, similarPairsName, lshSliceLengths, maxCheck, log2BucketCount):
            raise TypeError("findSimilarPairs7(): lshName, similarPairsName, lshSliceLengths, maxCheck and "
                            "log2BucketCount are required")
        lengths = np.ascontiguousarray([int(x) for x in lshSliceLengths], dtype=np.int32)
        capi.check(capi.load().em2_matrix_find_similar_pairs7(self._handle, _b(geneSetName), _b(cellSetName), _b(lshName),
                                                              _b(similarPairsName), k, similarityThreshold,
                                                              capi._ptr(lengths), len(lengths), maxCheck, log2BucketCount))

    # ---- src/PythonModule.cpp:926-934 ----
    def removeSimilarPairs(self, similarPairsName):
        capi.check(capi.load().em2_matrix_remove_similar_pairs(self._handle, _b(similarPairsName)))

    def _cell_set(self, cellSetName):
        lib = capi.load()
        count = ctypes.c_uint32(0)
        capi.check(lib.em2_matrix_cell_set(self._handle, _b(cellSetName), ctypes.byref(count), None))
        ids = np.zeros(count.value, dtype=np.uint32)
        capi.check(lib.em2_matrix_cell_set(self._handle, _b(cellSetName), ctypes.byref(count), capi._ptr(ids)))
        return ids

    # ---- src/PythonModule.cpp:1007-1034: the consumer of SimilarPairs ----
    def getCellGraphNames(self):
        return sorted(self._cellGraphs)          # std::map order (src/ExpressionMatrix.cpp:1782-1789)

    def createCellGraph(self, graphName=_REQUIRED, cellSetName="AllCells", similarPairsName=_REQUIRED,
                        similarityThreshold=0.5, k=20, keepIsolatedVertices=False):
        """ExpressionMatrix::createCellGraph (src/ExpressionMatrix.cpp:1795-1845).  The graph lives in memory, like
        the reference's; its edges are built on the GPU (em2_cell_graph_edges) in the reference's insertion order."""
        if graphName is _REQUIRED or similarPairsName is _REQUIRED:
            raise TypeError("createCellGraph(): graphName and similarPairsName are required")
        _b(graphName)
        if graphName in self._cellGraphs:
            raise RuntimeError("Graph " + graphName + " already exists.")
        try:
            graphCells = self._cell_set(cellSetName)
        except RuntimeError:
            raise RuntimeError("Cell set " + cellSetName + " does not exists.") from None   # sic, :1813
        _, _, _, similarPairsCellSetName = files.similar_pairs_info(self.directoryName, similarPairsName)
        storedK, pairs, used = files.read_similar_pairs(self.directoryName, similarPairsName)
        similarPairsCells = self._cell_set(similarPairsCellSetName)
        v0, v1, similarity = capi.cell_graph_edges(pairs, used, similarPairsCells, graphCells, similarityThreshold, k)
        vertices = graphCells
        edgeVertices = (v0, v1)
        isolatedRemoved = 0
        if not keepIsolatedVertices:                # CellGraph::removeIsolatedVertices (src/CellGraph.cpp:189-205)
            connected = np.zeros(len(graphCells), dtype=bool)
            connected[v0] = True
            connected[v1] = True
            isolatedRemoved = int(len(graphCells) - connected.sum())
            vertices = graphCells[connected]
            position = np.cumsum(connected, dtype=np.int64) - 1     # vertex index once the isolated ones are gone
            edgeVertices = (position[v0].astype(np.uint32), position[v1].astype(np.uint32))
        self._cellGraphs[graphName] = {
            "cellSetName": cellSetName, "similarPairsName": similarPairsName,
            "similarityThreshold": similarityThreshold, "maxConnectivity": k,
            "vertexCount": int(len(vertices)), "edgeCount": int(len(v0)),
            "isolatedRemovedVertexCount": isolatedRemoved,
            "vertexCellIds": vertices, "edgeCellIds": (graphCells[v0], graphCells[v1]), "edgeSimilarity": similarity,
            "edgeVertices": edgeVertices,
        }

    def _cell_graph(self, graphName):
        if graphName not in self._cellGraphs:
            raise RuntimeError("Graph " + graphName + " does not exist.")
        return self._cellGraphs[graphName]

    def getCellGraphEdges(self, graphName):
        """[(cellId0, cellId1)] per edge, in edge-list order (src/ExpressionMatrix.cpp:1892-1913)."""
        c0, c1 = self._cell_graph(graphName)["edgeCellIds"]
        return list(zip(c0.tolist(), c1.tolist()))

    def labelPropagationClustering(self, graphName, seed=231, stableIterationCountThreshold=3, maxIterationCount=100):
        """CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612), the first step of
        ExpressionMatrix::createClusterGraph (src/ExpressionMatrix.cpp:2145-2149; defaults from
        ClusterGraphCreationParameters, src/ClusterGraph.hpp:48-50).  The reference keeps the result in the
        clusterId of every graph vertex; here it is returned: (cellIds, clusterIds), one entry per vertex, clusters
        numbered from 0 by decreasing size.  The rest of createClusterGraph (ClusterGraph) is outside SURVEY.md 8."""
        g = self._cell_graph(graphName)
        v0, v1 = g["edgeVertices"]
        clusters, iterations = capi.cell_graph_label_propagation(g["vertexCellIds"], v0, v1, g["edgeSimilarity"], seed,
                                                                 stableIterationCountThreshold, maxIterationCount)
        g["clusterIds"] = clusters
        g["labelPropagationIterations"] = iterations
        return g["vertexCellIds"].copy(), clusters

    def _cell_graph_information(self, graphName):
        """The CellGraphInformation fields stored beside the graph (src/ExpressionMatrix.cpp:1824-1839); the
        reference shows them in its HTTP UI only, hence no public name here."""
        g = self._cell_graph(graphName)
        return {key: g[key] for key in ("cellSetName", "similarPairsName", "similarityThreshold", "maxConnectivity",
                                        "vertexCount", "edgeCount", "isolatedRemovedVertexCount")}

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

"""BASELINE.json's configurations at their full single-GPU sizes, through the C ABI, against the CPU oracle.

    configs[1]  100k cells x 20k genes, 1024-bit signatures, findSimilarPairs4
    configs[2]  1M cells x 30k genes, 1024-bit signatures, findSimilarPairs4 (the configuration the metric is quoted on)
    configs[3]  1M cells, 2048-bit signatures, findSimilarPairs5 (lshSliceLength 20, bucketOverflow 1000)
    configs[4]  1M cells: SimilarPairs (k=100) -> createCellGraph (k=20) -> label propagation

configs[1] is compared in full (every row against every column, the oracle's rows spread over the host threads).  The
oracle cannot run the 1M-cell ones in full (the pair loop is two hours of one core), so each test compares what the
oracle can do in seconds -- 10^4 sampled rows against ALL columns, sampled cells' signatures, sampled fsp5 cells spread over the
id range, and for configs[4] the WHOLE edge list and every label (the graph's oracle is linear in the pairs) -- bit for
bit, and adds size-independent properties over EVERY row of the result: the order the reference's sort leaves
(src/orderPairs.hpp:44-52), no self and no duplicate neighbours, value-initialised unused slots
(src/MemoryMappedVector.hpp:451-454), and on thousands of sampled entries that the stored float is
float(cos(pi * mismatches / lshCount)) of the two cells' actual signatures (src/Lsh.cpp:229-249,254-274)."""
import numpy as np
import pytest

from expressionmatrix2_amd import capi, sharded, synthetic

pytestmark = pytest.mark.gpu

K, THR, SEED = 100, 0.2, 231


def popcount_mismatches(sig, a, b):
    return np.bitwise_count(sig[a] ^ sig[b]).sum(axis=-1)


def check_structure_of_every_row(torch, pairs, used, k, cell_count):
    """pairs int32 [rows, k, 2] and used int32 [rows] on the device (row i = cell i)."""
    rows = pairs.shape[0]
    cell = pairs[:, :, 0].to(torch.int64) & 0xFFFFFFFF
    sim = pairs[:, :, 1].contiguous().view(torch.float32)
    u = used.to(torch.int64)
    assert int(u.max()) <= k and int(u.min()) >= 0
    live = torch.arange(k, device=pairs.device).unsqueeze(0) < u.unsqueeze(1)
    # unused slots are all-zero bytes
    assert not bool((pairs[:, :, 0][~live] != 0).any()) and not bool((pairs[:, :, 1][~live] != 0).any())
    # similarity descending, equal similarities by ascending cell id
    both = live[:, 1:]
    descending = sim[:, 1:] <= sim[:, :-1]
    tie_order = (sim[:, 1:] != sim[:, :-1]) | (cell[:, 1:] > cell[:, :-1])
    assert bool((descending | ~both).all()) and bool((tie_order | ~both).all())
    # neighbours are cells of the problem, never the cell itself, never twice
    assert bool(((cell < cell_count) | ~live).all())
    own = torch.arange(rows, device=pairs.device).unsqueeze(1)
    assert bool(((cell != own) | ~live).all())
    ordered, _ = torch.sort(torch.where(live, cell, torch.full_like(cell, -1) - torch.arange(k, device=pairs.device)), dim=1)
    assert bool((ordered[:, 1:] != ordered[:, :-1]).all())


def check_similarities_against_signatures(pairs_host, used_host, sig_host, lsh_count, thr, rows):
    """The stored similarity of sampled entries = float(similarityTable[mismatches of the two signatures]) > thr."""
    table = capi.similarity_table(lsh_count)
    for row in rows:
        n = int(used_host[row])
        if not n:
            continue
        others = pairs_host["cell"][row, :n]
        m = popcount_mismatches(sig_host, np.full(n, row), others)
        assert np.array_equal(pairs_host["similarity"][row, :n].view(np.uint32), table[m].astype(np.float32).view(np.uint32))
        assert (table[m] > thr).all()


def assert_rows_equal(pairs, used, cell, sim, oused):
    assert np.array_equal(used, oused)
    assert np.array_equal(pairs["cell"], cell)
    assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))


def run_fsp4_config(torch, oracle, cells, genes, lsh_count, sampled_rows=10240, signature_cells=64):
    """sampled_rows rows x ALL columns against the oracle (>= 10^4 rows at 1M cells; 0 = the whole result,
    what 8(d) asks of config B), the oracle's rows spread over the host's threads."""
    import bench
    device = torch.device("cuda", 0)
    pipe = sharded.DevicePipeline(cells, genes, lsh_count, K, THR, world_size=1, rank=0, dist=None, device=device)
    toc, data = synthetic.expression_shard(0, cells, genes, density=0.01, device=device)
    vectors_host = capi.lsh_generate_vectors(genes, lsh_count, SEED)
    pipe.set_inputs(toc, data, torch.from_numpy(vectors_host).to(device))
    pipe.step()
    pipe.check()
    torch.cuda.synchronize()
    sig_host = pipe.full_sig[:cells].cpu().numpy().view(np.uint64)

    # signatures of cells spread over the matrix (Lsh::computeCellLshSignatures, src/Lsh.cpp:118-224)
    toc_host = toc.cpu().numpy()
    picks = np.unique(np.linspace(0, cells - 1, signature_cells).astype(np.int64))
    for c in picks:
        t_h, g_h, c_h = synthetic.csr_to_host(toc[c:c + 2] - toc[c], data[int(toc_host[c]):int(toc_host[c + 1])])
        expect = oracle.compute_signatures(t_h, g_h, c_h, genes, vectors_host, lsh_count)
        assert np.array_equal(expect[0], sig_host[c]), "signature of cell %d" % c

    # sampled rows against all columns (findSimilarPairs4's per-cell contract, src/ExpressionMatrixLsh.cpp:200-285)
    ranges = bench.sample_ranges(pipe.owned_ranges(), sampled_rows)
    fetched = {r: pipe.results_for(*r) for r in ranges}
    checked = 0
    for begin, end, cell, sim, oused in bench.oracle_rows_parallel(oracle, sig_host, lsh_count, K, THR, ranges):
        r = next(r for r in ranges if r[0] <= begin and end <= r[1])
        got_pairs, got_used = fetched[r]
        assert_rows_equal(got_pairs[begin - r[0]:end - r[0]], got_used[begin - r[0]:end - r[0]], cell, sim, oused)
        checked += end - begin
    assert checked >= (sampled_rows or cells)

    check_structure_of_every_row(torch, pipe.pairs, pipe.used, K, cells)
    return pipe, sig_host


def test_configs1_100k_cells_20k_genes(oracle):
    import torch
    cells = 100000
    pipe, sig_host = run_fsp4_config(torch, oracle, cells, 20000, 1024, sampled_rows=0)          # EVERY row against the oracle
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 3           # the matrix-core form, as bench.py runs it
    pairs, used = pipe.results_for(0, cells)
    check_similarities_against_signatures(pairs, used, sig_host, 1024, THR, range(0, cells, 97))


@pytest.fixture(scope="module")
def million(oracle):
    """configs[2]: one step of the pipeline at 1M cells x 30k genes, checked; configs[4] continues from its SimilarPairs."""
    import torch
    pipe, sig_host = run_fsp4_config(torch, oracle, 1000000, 30000, 1024)
    return torch, pipe, sig_host


def test_configs2_1m_cells_30k_genes(million):
    torch, pipe, sig_host = million
    cells = pipe.cell_count
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 3
    rows = np.unique(np.concatenate([np.arange(0, cells, 4099), np.arange(cells - 64, cells)]))
    pairs = np.zeros((cells, K), dtype=capi.PAIR_DTYPE)
    raw = pipe.pairs.cpu().numpy().view(np.uint32)
    pairs["cell"], pairs["similarity"] = raw[:, :, 0], raw[:, :, 1].view(np.float32)
    check_similarities_against_signatures(pairs, pipe.used.cpu().numpy().view(np.uint32), sig_host, 1024, THR, rows)


@pytest.mark.parametrize("world,rank", [(8, 3), (4, 1), (2, 1), (8, 7)])
def test_configs2_row_shards_on_the_matrix_cores(oracle, million, world, rank):
    """BASELINE configs[2] names 8 GPUs with the rows split by contiguous id range: the one GPU of the box plays
    rank r of P -- em2_dev_find_similar_pairs4 with the rank's row range against all columns, which takes the rows form on the
    matrix cores (form 4).  EVERY row of the shard equals the one-GPU result, and 10 240 rows of it (16 places) the oracle's
    (src/ExpressionMatrixLsh.cpp:200-285, per-cell form)."""
    import bench
    torch, pipe, sig_host = million
    cells = pipe.cell_count
    begin, end = sharded.shard_range(cells, world, rank)
    rows = end - begin
    assert capi.dev_find_similar_pairs4_form_for(cells, rows, 1024) == 4
    ws_bytes = capi.dev_find_similar_pairs4_workspace(cells, rows, 1024, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pipe.device)
    pairs = torch.zeros((rows, K, 2), dtype=torch.int32, device=pipe.device)
    used = torch.zeros(rows, dtype=torch.int32, device=pipe.device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        capi.dev_find_similar_pairs4(pipe.full_sig.data_ptr(), cells, begin, end, 1024, K, THR, pairs.data_ptr(), used.data_ptr(),
                                     ws.data_ptr(), ws_bytes, stream)
    capi.dev_find_similar_pairs4_status(ws.data_ptr(), rows, K, stream)
    info = capi.dev_find_similar_pairs4_last_launch()
    assert info["form"] == 4 and info["matrix_pairs"] >= rows * float(cells & ~31)
    print("rows form, rank %d of %d: kernel %.1f ms at %.2f GHz" % (rank, world, info["matrix_kernel_ms"], info["matrix_clock_ghz"]))
    assert torch.equal(used, pipe.used[begin:end]) and torch.equal(pairs, pipe.pairs[begin:end])
    ranges = bench.sample_ranges([(begin, end)], 10240)
    host_pairs = pairs.cpu().numpy().view(np.uint32)
    host_used = used.cpu().numpy().view(np.uint32)
    checked = 0
    for b, e, cell, sim, oused in bench.oracle_rows_parallel(oracle, sig_host, 1024, K, THR, ranges):
        lo, hi = b - begin, e - begin
        assert np.array_equal(host_used[lo:hi], oused) and np.array_equal(host_pairs[lo:hi, :, 0], cell)
        assert np.array_equal(host_pairs[lo:hi, :, 1], sim.view(np.uint32))
        checked += e - b
    assert checked >= 10240


def test_configs4_1m_cells_graph_and_labels(oracle, million):
    """createCellGraph (src/CellGraph.cpp:33-117) and labelPropagationClustering (src/CellGraph.cpp:443-612) on the
    million-cell SimilarPairs: the WHOLE edge list against the oracle's add_edge order (hash-table form of the literal
    restatement, tests/test_cell_graph_cpu.py holds the two equal) and EVERY label against the oracle's serial run."""
    torch, pipe, sig_host = million
    cells, graph_k = pipe.cell_count, 20
    ids = np.arange(cells, dtype=np.uint32)
    device = pipe.pairs.device
    d_v0 = torch.empty(cells * graph_k, dtype=torch.int32, device=device)
    d_v1 = torch.empty(cells * graph_k, dtype=torch.int32, device=device)
    d_sim = torch.empty(cells * graph_k, dtype=torch.float32, device=device)
    edges = capi.dev_cell_graph_edges_to_device(pipe.pairs.data_ptr(), pipe.used.data_ptr(), cells, K, ids, ids, THR, graph_k,
                                                d_v0.data_ptr(), d_v1.data_ptr(), d_sim.data_ptr())
    clusters, iterations = capi.dev_cell_graph_label_propagation(ids, d_v0.data_ptr(), d_v1.data_ptr(), d_sim.data_ptr(), edges)
    v0 = d_v0[:edges].cpu().numpy().view(np.uint32)
    v1 = d_v1[:edges].cpu().numpy().view(np.uint32)
    sim = d_sim[:edges].cpu().numpy()
    raw = pipe.pairs.cpu().numpy().view(np.uint32)
    ev0, ev1, es = oracle.cell_graph_edges(raw[:, :, 0], raw[:, :, 1].view(np.float32), pipe.used.cpu().numpy().view(np.uint32),
                                           ids, ids, THR, graph_k, hashed=True)
    assert len(ev0) == edges and edges > 5 * cells
    assert np.array_equal(ev0, v0) and np.array_equal(ev1, v1) and np.array_equal(es.view(np.uint32), sim.view(np.uint32))
    oc, oit = oracle.label_propagation(ids, v0, v1, sim)
    assert oit == iterations
    assert np.array_equal(oc, clusters)
    # the clusters are the planted ones: 64 large groups (renumbered by decreasing size) hold nearly every cell
    sizes = np.bincount(clusters)
    assert sizes[:64].sum() > 0.99 * cells and (np.diff(sizes) <= 0).all()


def clustered_signatures_on_device(torch, cells, lsh_count, device, cluster_count=64, flip=0.15, seed=4321, chunk=65536):
    """bench.py's scan-only input: cluster centre with every bit flipped with probability `flip`."""
    import bench
    return bench.synthetic_signatures(torch, cells, lsh_count, device, cluster_count, flip, seed, chunk)


def test_configs3_1m_cells_2048_bits_fsp5(oracle):
    """findSimilarPairs5 (src/ExpressionMatrixLsh.cpp:312-501) at 1M cells x 2048 bits: 2048 cells in 16 places of the id range
    against the oracle, the structure of every row, and for sampled entries the similarity and the shared bucket."""
    import torch
    cells, L, q, overflow = 1000000, 2048, 20, 1000
    device = torch.device("cuda", 0)
    sig = clustered_signatures_on_device(torch, cells, L, device)
    pairs = torch.zeros((cells, K, 2), dtype=torch.int32, device=device)
    used = torch.zeros(cells, dtype=torch.int32, device=device)
    capi.dev_find_similar_pairs5(sig.data_ptr(), cells, 0, cells, L, K, THR, q, overflow, pairs.data_ptr(), used.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sig_host = sig.cpu().numpy().view(np.uint64)
    listed = np.unique(np.concatenate([np.arange(b, b + 128) for b in
                                       [(cells - 128) * i // 15 + (7 * i) % 64 * (0 < i < 15) for i in range(16)]])).astype(np.uint32)
    assert len(listed) >= 2000
    cell, sim, oused = oracle.find_similar_pairs5_cells(sig_host, L, K, THR, q, overflow, listed)
    raw = pairs.cpu().numpy().view(np.uint32)
    used_host = used.cpu().numpy().view(np.uint32)
    assert np.array_equal(used_host[listed], oused)
    assert np.array_equal(raw[listed][:, :, 0], cell)
    assert np.array_equal(raw[listed][:, :, 1], sim.view(np.uint32))
    check_structure_of_every_row(torch, pairs, used, K, cells)
    host = np.zeros((cells, K), dtype=capi.PAIR_DTYPE)
    host["cell"], host["similarity"] = raw[:, :, 0], raw[:, :, 1].view(np.float32)
    rows = np.arange(0, cells, 9973)
    check_similarities_against_signatures(host, used_host, sig_host, L, THR, rows)
    # every stored neighbour shares one of the 102 slice values with its cell (the candidate rule, :414-431)
    slices = L // q
    bits = np.unpackbits(sig_host[rows].astype(">u8").view(np.uint8).reshape(len(rows), -1), axis=1)[:, :slices * q]
    for i, row in enumerate(rows):
        n = int(used_host[row])
        if not n:
            continue
        other_bits = np.unpackbits(sig_host[host["cell"][row, :n]].astype(">u8").view(np.uint8).reshape(n, -1), axis=1)[:, :slices * q]
        same = (other_bits.reshape(n, slices, q) == bits[i].reshape(1, slices, q)).all(axis=2)
        assert same.any(axis=1).all()

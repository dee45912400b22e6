"""Row-sharded execution of the LSH similar-pairs path: one process per GPU, torch.distributed for the one
exchange step (all-gather of signature shards; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

SURVEY.md 8(e): cells are split into contiguous id ranges; every rank projects its own cells, the signature
shards are all-gathered (cell-major layout makes the gathered buffer the full signature array), and every
rank scans its own ROWS against all columns, which is exactly the per-cell contract of findSimilarPairs4 --
no collective on the result, no cross-rank merge of top-k state.

DevicePipeline (device-resident, used by bench.py) goes one step further when the problem is large enough: the
SHARDED SYMMETRIC scan (csrc/em2_scan_sharded.hip, include/em2_lsh.h: em2_dev_fsp4_sharded_*) evaluates every unordered
pair once across all ranks instead of once per rank and side.  64-cell blocks are dealt round-robin to the ranks;
four kernel phases are separated by two all_reduce(MAX) of a 4-byte-per-cell snapshot array and one all_gather of
the ranks' deferred-candidate pools.  EM2_SHARDED_SCAN=0 keeps the row-shard scan.

torch is used here for device memory, streams and the collective only; the compute goes through the C ABI
(capi.dev_*).  The compute callables can be replaced (the CPU tests inject the oracle, there being no GPU)."""
import os

import numpy as np

from . import capi


def shard_size(cell_count, world_size):
    return (cell_count + world_size - 1) // world_size


def shard_range(cell_count, world_size, rank):
    """Contiguous cell-id range [begin, end) owned by `rank`."""
    size = shard_size(cell_count, world_size)
    begin = min(cell_count, rank * size)
    end = min(cell_count, begin + size)
    return begin, end


def gather_signatures(local_signatures, cell_count, world_size, rank, dist=None):
    """All-gather of per-rank signature shards [rows_r, words] (int64 torch tensors) into the full
    [cell_count, words] array.  Shards are padded to the common shard size for the collective."""
    import torch
    words = local_signatures.shape[1]
    size = shard_size(cell_count, world_size)
    if world_size == 1:
        return local_signatures
    padded = local_signatures
    if local_signatures.shape[0] != size:
        padded = torch.zeros((size, words), dtype=local_signatures.dtype, device=local_signatures.device)
        padded[:local_signatures.shape[0]] = local_signatures
    full = torch.empty((size * world_size, words), dtype=local_signatures.dtype, device=local_signatures.device)
    dist.all_gather_into_tensor(full, padded.contiguous())
    return full[:cell_count]


class DevicePipeline:
    """Device-resident pipeline of one rank: CSR shard -> signatures -> all-gather -> row-shard scan.
    All buffers are allocated once in __init__; step() only enqueues kernels and the collective."""

    def __init__(self, cell_count, gene_count, lsh_count, k, similarity_threshold, world_size=1, rank=0,
                 dist=None, device="cuda"):
        import torch
        self.torch = torch
        self.cell_count = cell_count
        self.gene_count = gene_count
        self.lsh_count = lsh_count
        self.k = k
        self.thr = float(similarity_threshold)
        self.world_size = world_size
        self.rank = rank
        self.dist = dist
        self.device = device
        self.words = capi.word_count(lsh_count)
        self.row_begin, self.row_end = shard_range(cell_count, world_size, rank)
        self.rows = self.row_end - self.row_begin
        size = shard_size(cell_count, world_size)
        self.shard = size
        self.local_sig = torch.zeros((size, self.words), dtype=torch.int64, device=device)
        self.full_sig = (self.local_sig if world_size == 1 else
                         torch.empty((size * world_size, self.words), dtype=torch.int64, device=device))
        self.sharded = None
        # From EM2_SHARDED_MIN_CELLS cells on (default 100000): below that the fixed costs of the extra phases and
        # collectives outweigh the halved pair work.
        min_cells = int(os.environ.get("EM2_SHARDED_MIN_CELLS", "100000"))
        if world_size > 1 and k > 0 and cell_count >= min_cells and os.environ.get("EM2_SHARDED_SCAN", "1") != "0":
            plan = capi.dev_fsp4_sharded_plan(cell_count, lsh_count, k, rank, world_size)
            if plan["eligible"]:
                self.sharded = plan
        self.pairs = self.used = self.scan_ws = None
        self.scan_ws_bytes = 0
        self.scan_error = None
        if self.sharded:
            plan = self.sharded
            # results indexed by GLOBAL cell id; this rank fills the rows of its own blocks (rank, rank+world, ...)
            self.global_pairs = torch.zeros((cell_count, k, 2), dtype=torch.int32, device=device)
            self.global_used = torch.zeros(cell_count, dtype=torch.int32, device=device)
            self.shard_ws = torch.empty(plan["workspace_bytes"] + 256, dtype=torch.uint8, device=device)
            skip = (-self.shard_ws.data_ptr()) % 256
            self.shard_ws = self.shard_ws[skip:skip + plan["workspace_bytes"]]
            self.snap = self.shard_ws[plan["snap_offset"]:plan["snap_offset"] + 4 * cell_count].view(torch.int32)
            self.pool = self.shard_ws[plan["pool_offset"]:plan["pool_offset"] + 8 * plan["pool_capacity"]].view(torch.int64)
            self.gathered = self.shard_ws[plan["gathered_offset"]:
                                          plan["gathered_offset"] + 8 * plan["gathered_capacity"]].view(torch.int64)
            self.sorted_area = self.shard_ws[plan["sorted_offset"]:
                                             plan["sorted_offset"] + 8 * plan["gathered_capacity"]].view(torch.int64)
            self.count_buf = torch.zeros(2, dtype=torch.int64, device=device)
            # entries travel by all_to_all (each rank receives only the candidates of its own cells) when the owner
            # of a target cell is a bit field of the entry key (a power-of-two world); otherwise the pools are all_gathered
            self.exchange_all_to_all = (world_size & (world_size - 1)) == 0
        else:
            self._allocate_row_shard_scan()
        self.proj_ws_bytes = capi.dev_compute_signatures_workspace(max(1, self.rows), lsh_count)
        self.proj_ws = torch.empty(self.proj_ws_bytes, dtype=torch.uint8, device=device)
        self.vector_aux = torch.empty(capi.dev_vector_aux_bytes(gene_count, lsh_count), dtype=torch.uint8,
                                      device=device)
        self.scan_events = []
        self.timing = None          # dict name -> [ms, ...] while bench.py's diagnostic leg runs (synchronises at every mark)
        self._mark_time = None

    def start_timing(self):
        """Diagnostics: from now on every stage of step() ends with a device synchronisation and its wall time is appended to
        self.timing[name] (NOT for measured runs: the stages no longer overlap)."""
        import time
        self.timing = {}
        self.torch.cuda.synchronize()
        self._mark_time = time.perf_counter()

    def stop_timing(self):
        timing, self.timing = self.timing, None
        return {name: sum(values) / len(values) for name, values in (timing or {}).items()}

    def _mark(self, name):
        if self.timing is None:
            return
        import time
        self.torch.cuda.synchronize()
        now = time.perf_counter()
        self.timing.setdefault(name, []).append((now - self._mark_time) * 1e3)
        self._mark_time = now

    def _allocate_row_shard_scan(self):
        torch = self.torch
        if self.scan_ws is not None:
            return
        self.pairs = torch.zeros((max(1, self.rows), max(1, self.k), 2), dtype=torch.int32, device=self.device)
        self.used = torch.zeros(max(1, self.rows), dtype=torch.int32, device=self.device)
        self.scan_ws_bytes = capi.dev_find_similar_pairs4_workspace(self.cell_count, self.rows, self.lsh_count, self.k)
        self.scan_ws = torch.empty(max(1, self.scan_ws_bytes), dtype=torch.uint8, device=self.device)

    def set_inputs(self, toc, data, vectors):
        """toc int64 [rows+1] (relative to this shard), data int64-viewed em2_count [nnz], vectors float64
        [gene_count, lsh_count]; all on the device."""
        self.toc, self.data, self.vectors = toc, data, vectors
        stream = self.torch.cuda.current_stream().cuda_stream
        capi.dev_prepare_vectors(vectors.data_ptr(), self.gene_count, self.lsh_count, self.vector_aux.data_ptr(),
                                 stream)

    def project(self):
        stream = self.torch.cuda.current_stream().cuda_stream
        if self.rows:
            capi.dev_compute_signatures(self.toc.data_ptr(), self.data.data_ptr(), self.rows, self.gene_count,
                                        self.vectors.data_ptr(), self.vector_aux.data_ptr(), self.lsh_count,
                                        self.local_sig.data_ptr(), self.proj_ws.data_ptr(), self.proj_ws_bytes,
                                        stream)
        self._mark("projection")

    def projection_tier(self):
        """Which first tier the last project() ran (waits for the device): capi.TIER_NAMES."""
        if not self.rows:
            return None
        return capi.dev_compute_signatures_tier(self.proj_ws.data_ptr(), self.rows, self.lsh_count, True)

    def exchange(self):
        if self.world_size > 1:
            self.dist.all_gather_into_tensor(self.full_sig, self.local_sig)
        self._mark("all_gather_signatures")

    def scan(self, record_events=False):
        torch = self.torch
        if record_events:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.sharded:
            # A pool overflow, a hand-off time-out or an exception in one of this rank's first three phases is folded into
            # the flag the ranks reduce before the exchange (_scan_sharded keeps issuing the collectives until then), so all
            # ranks leave the sharded form TOGETHER, for this and all later steps.  An exception after that agreement is not
            # caught: the process ends, and the launcher ends the job (a rank that skipped collectives must not go on).
            if not self._scan_sharded():
                self.sharded = None
                self._allocate_row_shard_scan()
        if not self.sharded and self.rows:
            stream = torch.cuda.current_stream().cuda_stream
            capi.dev_find_similar_pairs4(self.full_sig.data_ptr(), self.cell_count, self.row_begin, self.row_end,
                                         self.lsh_count, self.k, self.thr, self.pairs.data_ptr(),
                                         self.used.data_ptr(), self.scan_ws.data_ptr(), self.scan_ws_bytes, stream)
        self._mark("scan_rows" if not self.sharded else "scan_tail")
        if record_events:
            e1.record()
            self.scan_events.append((e0, e1))

    def _scan_sharded(self):
        """The four phases of the sharded symmetric scan with their collectives; False if a pool overflowed."""
        torch, dist, plan = self.torch, self.dist, self.sharded
        stream = torch.cuda.current_stream().cuda_stream
        world = self.world_size

        def phase(number, gathered_count=0):
            capi.dev_fsp4_sharded_phase(number, self.full_sig.data_ptr(), self.cell_count, self.lsh_count, self.k, self.thr,
                                        self.rank, world, self.global_pairs.data_ptr(), self.global_used.data_ptr(),
                                        self.shard_ws.data_ptr(), plan["workspace_bytes"], gathered_count, stream)

        broken = None

        def guarded(number):
            nonlocal broken
            if broken is None:
                try:
                    phase(number)
                except (RuntimeError, TypeError, ValueError) as error:
                    broken = error

        guarded(0)
        self._mark("phase0_prefix_full_rows")
        dist.all_reduce(self.snap, op=dist.ReduceOp.MAX)
        self._mark("all_reduce_snapshots")
        guarded(1)
        self._mark("phase1_prefix_columns")
        dist.all_reduce(self.snap, op=dist.ReduceOp.MAX)
        self._mark("all_reduce_snapshots")
        guarded(2)
        used, overflow = 0, 1
        if broken is None:
            try:
                used, overflow = capi.dev_fsp4_sharded_status(self.cell_count, self.k, self.rank, world,
                                                              self.shard_ws.data_ptr(), stream)
            except RuntimeError as error:            # a hand-off timed out: keep the collectives in step, report in check()
                self.scan_error = error
        else:
            import sys
            print("[em2] sharded symmetric scan failed on rank %d (%s); all ranks use row shards" % (self.rank, broken),
                  file=sys.stderr)
        self._mark("phase2_tiles")
        self.count_buf[0] = used
        self.count_buf[1] = overflow
        dist.all_reduce(self.count_buf, op=dist.ReduceOp.MAX)
        max_used, any_overflow = (int(x) for x in self.count_buf.cpu())         # the one agreement point of the step
        self._mark("agree_counts")
        if any_overflow:
            return False
        if self.exchange_all_to_all:
            received = self._exchange_all_to_all(used, phase)
            self._mark("exchange_candidates")
            phase(3, received)
            self._mark("phase3_sort_replay")
            return True
        if max_used:
            if used < max_used:
                self.pool[used:max_used].fill_(-1)           # ~0: sentinels sort behind every real entry
            dist.all_gather_into_tensor(self.gathered[:world * max_used], self.pool[:max_used])
        self._mark("exchange_candidates")
        phase(3, world * max_used)
        self._mark("phase3_sort_replay")
        return True

    def _exchange_all_to_all(self, used, phase):
        """Pool entries grouped by owner rank (phase 4) -> all_to_all_single -> self.gathered; returns the count received."""
        torch, dist, plan = self.torch, self.dist, self.sharded
        world = self.world_size
        phase(4, used)
        send = self.sorted_area[:used]
        owners = (send >> plan["owner_shift"]) & (world - 1)
        send_counts = torch.bincount(owners, minlength=world) if used else torch.zeros(world, dtype=torch.int64, device=send.device)
        staged = dist.get_backend() != "nccl"               # gloo has no all_to_all on device tensors: go through the host
        # every rank's counts to every rank in ONE collective and ONE read-back (row r = what rank r sends to each rank)
        counts_in = send_counts.cpu() if staged else send_counts
        matrix = torch.empty(world * world, dtype=torch.int64, device=counts_in.device)
        dist.all_gather_into_tensor(matrix, counts_in)
        matrix = matrix.cpu().view(world, world)
        send_list = [int(x) for x in matrix[self.rank]]
        recv_list = [int(x) for x in matrix[:, self.rank]]
        received = sum(recv_list)
        if received > plan["gathered_capacity"]:
            raise RuntimeError("sharded scan: received more entries than the exchange area holds")
        if staged:
            recv_host = torch.empty(received, dtype=torch.int64)
            dist.all_to_all_single(recv_host, send.cpu(), recv_list, send_list)
            self.gathered[:received].copy_(recv_host)
        else:
            dist.all_to_all_single(self.gathered[:received], send, recv_list, send_list)
        return received

    def owned_ranges(self):
        """Global [begin, end) cell ranges whose results this rank holds after scan()."""
        if self.sharded:
            blocks = range(self.rank, self.sharded["blocks"], self.world_size)
            return [(64 * b, min(self.cell_count, 64 * b + 64)) for b in blocks]
        return [(self.row_begin, self.row_end)] if self.rows else []

    def results_for(self, begin, end):
        """(pairs[end-begin, k] of capi.PAIR_DTYPE, used[end-begin]) of owned global rows [begin, end), on the host."""
        if self.sharded:
            p, u = self.global_pairs[begin:end], self.global_used[begin:end]
        else:
            p, u = self.pairs[begin - self.row_begin:end - self.row_begin], self.used[begin - self.row_begin:end - self.row_begin]
        p = p.cpu().numpy().view(np.uint32)
        pairs = np.zeros((end - begin, self.k), dtype=capi.PAIR_DTYPE)
        pairs["cell"] = p[:, :, 0]
        pairs["similarity"] = p[:, :, 1].view(np.float32)
        return pairs, u.cpu().numpy().view(np.uint32)

    def step(self, record_events=False):
        self.project()
        self.exchange()
        self.scan(record_events)

    def check(self):
        """Synchronise and raise if the last scan did not complete (capi.dev_find_similar_pairs4_status)."""
        if self.scan_error is not None:
            raise self.scan_error
        if self.sharded:
            self.torch.cuda.synchronize()
        elif self.rows and self.k:
            capi.dev_find_similar_pairs4_status(self.scan_ws.data_ptr(), self.rows, self.k,
                                                self.torch.cuda.current_stream().cuda_stream)

    def results(self):
        """(pairs[rows,k] of capi.PAIR_DTYPE, used[rows]) for the contiguous row shard (row-shard scan only)."""
        if self.sharded:
            raise RuntimeError("results(): the sharded symmetric scan owns blocks, use owned_ranges() / results_for()")
        return self.results_for(self.row_begin, self.row_end)


# ---------------------------------------------------------------------------------------------------------------
# Collective form of ExpressionMatrix.findSimilarPairs4: every rank calls it with the same arguments.
# ---------------------------------------------------------------------------------------------------------------

class HipBackend:
    """The compute of one rank, through the C ABI on the current HIP device."""
    comm_device = "cuda"

    def project(self, toc, data, gene_count, vectors, lsh_count):
        return capi.compute_signatures(toc, data, gene_count, vectors, lsh_count)

    def scan_rows(self, signatures, row_begin, row_end, lsh_count, k, thr):
        import torch
        cell_count = signatures.shape[0]
        rows = row_end - row_begin
        d_sig = torch.from_numpy(np.ascontiguousarray(signatures).view(np.int64)).cuda()
        ws_bytes = capi.dev_find_similar_pairs4_workspace(cell_count, rows, lsh_count, k)
        ws = torch.empty(max(1, ws_bytes), dtype=torch.uint8, device="cuda")
        d_pairs = torch.zeros((max(1, rows), max(1, k), 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(max(1, rows), dtype=torch.int32, device="cuda")
        if rows:
            capi.dev_find_similar_pairs4(d_sig.data_ptr(), cell_count, row_begin, row_end, lsh_count, k, thr,
                                         d_pairs.data_ptr(), d_used.data_ptr(), ws.data_ptr(), ws_bytes,
                                         torch.cuda.current_stream().cuda_stream)
        if rows and k:
            capi.dev_find_similar_pairs4_status(ws.data_ptr(), rows, k, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        p = d_pairs[:rows].cpu().numpy().view(np.uint32)
        pairs = np.zeros((rows, k), dtype=capi.PAIR_DTYPE)
        if k:
            pairs["cell"] = p[:, :k, 0]
            pairs["similarity"] = p[:, :k, 1].view(np.float32)
        return pairs, d_used[:rows].cpu().numpy().view(np.uint32).copy()


    def bucket_rows(self, signatures, row_begin, row_end, lsh_count, k, thr, slice_length, bucket_overflow):
        import torch
        cell_count = signatures.shape[0]
        rows = row_end - row_begin
        d_sig = torch.from_numpy(np.ascontiguousarray(signatures).view(np.int64)).cuda()
        d_pairs = torch.zeros((max(1, rows), max(1, k), 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(max(1, rows), dtype=torch.int32, device="cuda")
        if rows:
            capi.dev_find_similar_pairs5(d_sig.data_ptr(), cell_count, row_begin, row_end, lsh_count, k, thr,
                                         slice_length, bucket_overflow, d_pairs.data_ptr(), d_used.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        p = d_pairs[:rows].cpu().numpy().view(np.uint32)
        pairs = np.zeros((rows, k), dtype=capi.PAIR_DTYPE)
        if k:
            pairs["cell"] = p[:, :k, 0]
            pairs["similarity"] = p[:, :k, 1].view(np.float32)
        return pairs, d_used[:rows].cpu().numpy().view(np.uint32).copy()


def _collect_and_write(matrix, similar_pairs_name, gene_set_name, cell_set_name, k, cell_count, begin, end, pairs,
                       used, dist, comm_device):
    """All-gather the row shards (padded to the common shard size); rank 0 writes SimilarPairs-<name>-*."""
    import torch
    from . import files
    world = dist.get_world_size()
    size = shard_size(cell_count, world)
    send_pairs = np.zeros((size, max(1, k), 2), dtype=np.int32)
    send_used = np.zeros(size, dtype=np.int32)
    if end > begin and k:
        send_pairs[:end - begin, :k, 0] = pairs["cell"].view(np.int32)
        send_pairs[:end - begin, :k, 1] = pairs["similarity"].view(np.int32)
    send_used[:end - begin] = used.view(np.int32)
    tp = torch.from_numpy(send_pairs).to(comm_device)
    tu = torch.from_numpy(send_used).to(comm_device)
    all_pairs = torch.empty((world * size,) + tuple(tp.shape[1:]), dtype=tp.dtype, device=tp.device)
    all_used = torch.empty(world * size, dtype=tu.dtype, device=tu.device)
    dist.all_gather_into_tensor(all_pairs, tp)
    dist.all_gather_into_tensor(all_used, tu)
    if dist.get_rank() == 0:
        ap = all_pairs[:cell_count].cpu().numpy().view(np.uint32)
        out = np.zeros((cell_count, k), dtype=capi.PAIR_DTYPE)
        if k:
            out["cell"] = ap[:, :k, 0]
            out["similarity"] = ap[:, :k, 1].view(np.float32)
        files.write_similar_pairs(matrix.directoryName, similar_pairs_name, gene_set_name, cell_set_name, k, out,
                                  all_used[:cell_count].cpu().numpy().view(np.uint32))
    dist.barrier()


def find_similar_pairs5_collective(matrix, gene_set_name, cell_set_name, lsh_name, similar_pairs_name, k,
                                   similarity_threshold, lsh_slice_length, bucket_overflow, dist, backend=None):
    """ExpressionMatrix::findSimilarPairs5 (src/ExpressionMatrixLsh.cpp:312-501) across the ranks of `dist`: every
    rank reads the stored signatures and builds the (replicated) bucket tables, queries only its own cell-id range
    (SURVEY.md 8(e)); rank 0 collects and writes.  No collective on the data path besides the result gather."""
    from . import files
    backend = backend or HipBackend()
    world = dist.get_world_size()
    rank = dist.get_rank()
    matrix._subset_sizes(gene_set_name, cell_set_name)          # the reference's lookup / emptiness errors (:326-345)
    lsh_count, signatures = files.read_lsh(matrix.directoryName, lsh_name)
    cell_count = matrix._subset_sizes(gene_set_name, cell_set_name)[1]
    if signatures.shape[0] != cell_count:
        raise RuntimeError("LSH object %s has a number of cells inconsistent with cell set %s" % (lsh_name, cell_set_name))
    if lsh_slice_length == 0:
        raise RuntimeError("findSimilarPairs5: lshSliceLength must be positive.")
    begin, end = shard_range(cell_count, world, rank)
    pairs, used = backend.bucket_rows(signatures, begin, end, lsh_count, k, similarity_threshold, lsh_slice_length,
                                      bucket_overflow)
    _collect_and_write(matrix, similar_pairs_name, gene_set_name, cell_set_name, k, cell_count, begin, end, pairs,
                       used, dist, backend.comm_device)
    return None


def find_similar_pairs4_collective(matrix, gene_set_name, cell_set_name, similar_pairs_name, k,
                                   similarity_threshold, lsh_count, seed, dist, backend=None):
    """ExpressionMatrix::findSimilarPairs4 (src/ExpressionMatrixLsh.cpp:155-290) across the ranks of `dist`:
    subset -> hyperplanes (same seed on every rank) -> each rank projects its cells -> all-gather -> each rank
    scans its rows -> rank 0 collects the rows and writes SimilarPairs-<name>-*.  Returns None like the reference."""
    import torch
    from . import files
    backend = backend or HipBackend()
    world = dist.get_world_size()
    rank = dist.get_rank()
    gene_count, toc, data = matrix._subset(gene_set_name, cell_set_name)      # raises the reference's errors
    cell_count = len(toc) - 1
    begin, end = shard_range(cell_count, world, rank)
    local_toc = (toc[begin:end + 1] - toc[begin]).astype(np.uint64)
    local_data = data[int(toc[begin]):int(toc[end])]
    vectors = capi.lsh_generate_vectors(gene_count, lsh_count, seed)
    words = capi.word_count(lsh_count)
    if end > begin:
        local_sig = backend.project(local_toc, local_data, gene_count, vectors, lsh_count)
    else:
        local_sig = np.zeros((0, words), dtype=np.uint64)
    del vectors
    t_local = torch.from_numpy(np.ascontiguousarray(local_sig).view(np.int64)).to(backend.comm_device)
    t_full = gather_signatures(t_local, cell_count, world, rank, dist)
    signatures = t_full.cpu().numpy().view(np.uint64)
    pairs, used = backend.scan_rows(signatures, begin, end, lsh_count, k, similarity_threshold)

    _collect_and_write(matrix, similar_pairs_name, gene_set_name, cell_set_name, k, cell_count, begin, end, pairs,
                       used, dist, backend.comm_device)
    return None

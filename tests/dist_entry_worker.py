"""Worker of tests/test_gpu_dist_entry.py: one rank of em2_dist_find_similar_pairs4_with.  All ranks share the one GPU of the
test box, so RCCL cannot carry the collectives (it refuses two ranks on one device); the transport table is filled with
ctypes callbacks that stage through the host and torch.distributed's gloo backend.  What is under test is the C entry's
choreography: every rank must end with the oracle's SimilarPairs rows of its contiguous range.  Prints one JSON line."""
import ctypes
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_binding
import synth
from expressionmatrix2_amd import capi


def main():
    cells, L, k, thr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    calls = {"all_gather": 0, "all_reduce": 0, "all_to_all": 0}

    def to_host(pointer, nbytes, stream):
        out = np.empty(nbytes, dtype=np.uint8)
        assert hip.hipStreamSynchronize(stream) == 0
        if nbytes:
            assert hip.hipMemcpy(out.ctypes.data, pointer, nbytes, 2) == 0
        return out

    def to_device(pointer, array):
        if array.nbytes:
            assert hip.hipMemcpy(pointer, array.ctypes.data, array.nbytes, 1) == 0

    def all_gather(context, send, recv, nbytes, stream):
        calls["all_gather"] += 1
        mine = torch.from_numpy(to_host(send, nbytes, stream))
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, mine)
        to_device(recv, torch.cat(parts).numpy())
        return 0

    def all_reduce_max_i32(context, buffer, count, stream):
        calls["all_reduce"] += 1
        values = torch.from_numpy(to_host(buffer, 4 * count, stream).view(np.int32))
        dist.all_reduce(values, op=dist.ReduceOp.MAX)
        to_device(buffer, values.numpy())
        return 0

    def all_to_all_v(context, send, send_bytes, send_offsets, recv, recv_bytes, recv_offsets, stream):
        calls["all_to_all"] += 1
        total = max(send_offsets[p] + send_bytes[p] for p in range(world))
        mine = to_host(send, total, stream)
        # gloo has no all_to_all: every rank publishes what it sends to everybody, the receivers pick their parts
        sizes = torch.tensor([send_bytes[p] for p in range(world)], dtype=torch.int64)
        all_sizes = [torch.empty(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        for source in range(world):
            n = int(all_sizes[source].sum())
            payload = torch.empty(n, dtype=torch.uint8)
            if source == rank:
                payload = torch.from_numpy(np.concatenate([mine[send_offsets[p]:send_offsets[p] + send_bytes[p]] for p in range(world)])
                                           if n else np.empty(0, dtype=np.uint8))
            dist.broadcast(payload, src=source)
            begin = int(all_sizes[source][:rank].sum())
            part = payload[begin:begin + int(all_sizes[source][rank])].numpy()
            assert len(part) == recv_bytes[source]
            if len(part):
                assert hip.hipMemcpy((recv or 0) + recv_offsets[source], part.ctypes.data, len(part), 1) == 0
        return 0

    table = capi.Collectives(None, world, rank, capi.ALL_GATHER_FN(all_gather), capi.ALL_REDUCE_MAX_I32_FN(all_reduce_max_i32),
                             capi.ALL_TO_ALL_V_FN(all_to_all_v))

    sig = synth.clustered_signatures(cells, L, cluster_count=9, flip=0.12, seed=cells + k)
    words = sig.shape[1]
    shard = (cells + world - 1) // world
    begin, end = min(cells, rank * shard), min(cells, rank * shard + shard)
    local = torch.zeros((shard, words), dtype=torch.int64, device="cuda")
    local[:end - begin] = torch.from_numpy(sig[begin:end].view(np.int64).copy()).cuda()
    everything = torch.zeros((shard * world, words), dtype=torch.int64, device="cuda")
    pairs = torch.zeros((max(1, end - begin), max(1, k), 2), dtype=torch.int32, device="cuda")
    used = torch.zeros(max(1, end - begin), dtype=torch.int32, device="cuda")
    ws_bytes = capi.dist_find_similar_pairs4_workspace(cells, L, k, rank, world)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    form = capi.dist_find_similar_pairs4_form(cells, L, k, world)
    stages = None
    # EM2_TEST_EXPECT_FAILURE_ON_RANK (with the diagnostic library's fault injection): that rank's call must raise, every
    # rank must come back (nobody may be left waiting in a collective), and the other ranks must still hold the right rows --
    # unless EM2_TEST_ALL_RANKS_FAIL, i.e. when the failure comes after the ranks' agreement: then the failing rank's rows
    # have travelled, and EVERY rank must report an error (the outcome is collective) instead of returning them
    failing = int(os.environ.get("EM2_TEST_EXPECT_FAILURE_ON_RANK", "-1"))
    all_fail = os.environ.get("EM2_TEST_ALL_RANKS_FAIL") == "1"
    for timed in (False, True):
        try:
            stages = capi.dist_find_similar_pairs4(table, local.data_ptr(), cells, L, k, thr, everything.data_ptr(), pairs.data_ptr(),
                                                   used.data_ptr(), ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream,
                                                   timed=timed)
            if rank == failing or (failing >= 0 and all_fail):
                print("rank %d: the injected failure was not reported" % rank, file=sys.stderr)
                sys.exit(4)
        except RuntimeError as error:
            if rank == failing and "injected failure" in str(error):
                stages = {}
            elif failing >= 0 and all_fail and "another rank failed" in str(error):
                stages = {}
            else:
                raise
        torch.cuda.synchronize()
        if rank == failing or (failing >= 0 and all_fail):
            continue
        oracle = oracle_binding.load_oracle()
        cell, sim, oused = oracle.find_similar_pairs4_rows(sig, L, k, thr, begin, end)
        got = pairs[:end - begin].cpu().numpy().view(np.uint32)
        ok = (np.array_equal(used[:end - begin].cpu().numpy().view(np.uint32), oused) and np.array_equal(got[:, :, 0], cell) and
              np.array_equal(got[:, :, 1], sim.view(np.uint32)) and
              np.array_equal(everything[:cells].cpu().numpy().view(np.uint64), sig))
        if not ok:
            print("PARITY FAILURE on rank %d (timed=%s)" % (rank, timed), file=sys.stderr)
            sys.exit(3)
    flags = torch.tensor([1], dtype=torch.int64)
    dist.all_reduce(flags)
    if rank == 0:
        print(json.dumps({"ranks_ok": int(flags[0]), "form": form, "calls": calls, "rows": end - begin, "stages_ms": stages}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

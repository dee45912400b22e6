#!/usr/bin/env python3
"""What a TILE-wide column bound would cost the matrix scan in records.

The step tests every result against min(row bound, column bound of ITS column): two vector instructions per result.  With one
bound per 32-column tile (the loosest of its columns) the min moves out of the per-result work.  That is a superset test --
the replay filters exactly -- but every extra pass is a record.  This script runs the bench's default workload once, takes the
FINAL bound of every cell (the mismatch count of its k-th neighbour, or the threshold's when the list is not full) and counts,
for sampled rows against all columns, the pairs that pass with per-column bounds and with per-tile bounds.

    python3 tools/analyze_tile_bound.py [cells]            (needs the GPU)
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from expressionmatrix2_amd import capi, sharded, synthetic          # noqa: E402


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    G, L, k, thr = 30000, 1024, 100, 0.2
    device = torch.device("cuda", 0)
    capi.load()
    vectors = torch.from_numpy(capi.lsh_generate_vectors(G, L, 231)).to(device)
    pipe = sharded.DevicePipeline(C, G, L, k, thr, device=device)
    toc, data = synthetic.expression_shard(pipe.row_begin, pipe.row_end, G, density=0.01, device=device)
    pipe.set_inputs(toc, data, vectors)
    pipe.step()
    torch.cuda.synchronize()
    pairs, used = pipe.pairs[:C], pipe.used[:C]
    last = pairs[torch.arange(C, device=device), (used.long() - 1).clamp(min=0), 1].view(torch.float32)
    m_last = torch.round(torch.acos(last.double().clamp(-1, 1)) * L / math.pi).to(torch.int32)
    m_thr = int(math.floor(math.acos(thr) * L / math.pi))
    bound = torch.where(used >= k, m_last, torch.full_like(m_last, m_thr))
    b = bound.float()
    print("cells %d, full lists %.3f, bound: mean %.1f sd %.1f min %d max %d; threshold's bound %d" %
          (C, float((used >= k).float().mean()), float(b.mean()), float(b.std()), int(bound.min()), int(bound.max()), m_thr))
    tiles = C // 32
    tile_max = bound[:tiles * 32].view(tiles, 32).max(dim=1).values
    spread = (tile_max.repeat_interleave(32) - bound[:tiles * 32]).float()
    print("tile max - own bound: mean %.1f, median %.1f, 90%% %.1f" % (float(spread.mean()), float(spread.median()), float(spread.quantile(0.9))))
    # signatures as +-1 halves
    sig = pipe.full_sig[:C].contiguous().view(torch.int64).view(C, -1)
    shifts = torch.arange(63, -1, -1, device=device, dtype=torch.int64)
    rows = torch.randperm(C, device=device)[:512]

    def unpack(words):
        bits = ((words.unsqueeze(-1) >> shifts) & 1).reshape(words.shape[0], -1)[:, :L]
        return (bits * 2 - 1).to(torch.float16)

    row_pm = unpack(sig[rows])
    own = column = tile = tile_only = 0
    chunk = 32 * 4096
    col_tile_bound = tile_max.repeat_interleave(32)
    for start in range(0, tiles * 32, chunk):
        end = min(tiles * 32, start + chunk)
        dot = row_pm @ unpack(sig[start:end]).T
        m = ((L - dot.float()) / 2).to(torch.int32)
        later = torch.arange(start, end, device=device).unsqueeze(0) > rows.unsqueeze(1)      # the symmetric scan: columns above the row
        mr = bound[rows].unsqueeze(1)
        own += int(((m <= mr) & later).sum())
        column += int(((m <= torch.maximum(mr, bound[start:end].unsqueeze(0))) & later).sum())
        tile += int(((m <= torch.maximum(mr, col_tile_bound[start:end].unsqueeze(0))) & later).sum())
    pairs_seen = float((tiles * 32 - rows.float()).clamp(min=0).sum())
    per_wave_tile = 2048.0 / pairs_seen
    print("passes per 64 x 32 wave-tile at the FINAL bounds: row side only %.2f, min(row, column) %.2f, min(row, tile) %.2f" %
          (own * per_wave_tile, column * per_wave_tile, tile * per_wave_tile))


if __name__ == "__main__":
    main()

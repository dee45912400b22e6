"""Debugging aid: bench-like data through the emulated sharded scan, the walk under test against the trusted one,
repeated; prints which (row block, column tile) pairs differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from expressionmatrix2_amd import capi, synthetic
n, G, L, k, thr = int(sys.argv[1]), 3000, 1024, 100, 0.2
toc, data = synthetic.expression_shard(0, n, G, density=0.01, device="cuda")
t, g, c = synthetic.csr_to_host(toc, data)
vectors = capi.lsh_generate_vectors(G, L, 231)
sig = capi.compute_signatures(t, capi.make_counts(g, c), G, vectors, L)
base = dict(item.split("=") for item in sys.argv[2].split(","))
os.environ.update(base)
os.environ["EM2_MATRIX_WALK"] = "0"
ref_pairs, ref_used = capi.find_similar_pairs4(sig, L, k, thr)
prefix = None
os.environ["EM2_MATRIX_WALK"] = sys.argv[3]
bits = np.unpackbits(sig.view(np.uint8), axis=1)
for attempt in range(int(sys.argv[4])):
    pairs, used = capi.find_similar_pairs4(sig, L, k, thr)
    bad = np.nonzero((pairs["cell"] != ref_pairs["cell"]).any(axis=1) | (pairs["similarity"].view(np.uint32) != ref_pairs["similarity"].view(np.uint32)).any(axis=1) | (used != ref_used))[0]
    if not len(bad):
        continue
    # pairs (row, other) whose stored similarity is wrong
    wrong = set()
    for r in bad[:200]:
        for i in range(used[r]):
            o = int(pairs["cell"][r][i])
            m = int((bits[r] != bits[o]).sum())
            if abs(float(pairs["similarity"][r][i]) - float(np.float32(np.cos(m * np.pi / L)))) > 1e-7:
                wrong.add((max(r, o), min(r, o), m, round(float(np.arccos(np.clip(pairs["similarity"][r][i], -1, 1)) * L / np.pi))))
    tiles = sorted(set((row // 64, col // 32) for row, col, _, _ in wrong))
    print("attempt", attempt, "bad rows", len(bad), "wrong pairs", len(wrong), "(row block, wave, column tile, tile parity):",
          [(rb, rb % 4, ct, ct & 1) for rb, ct in tiles][:8], "m true/stored", [(m, s) for _, _, m, s in sorted(wrong)][:6], flush=True)

"""Deterministic synthetic inputs shared by the tests and bench.py.

Everything is derived from a counter-based SplitMix64 hash implemented with numpy uint64 arithmetic, so the
same (seed, shape) gives the same bytes on any machine and numpy version."""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    """Vectorised SplitMix64 finaliser over uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def hash_u64(seed, *coords):
    """Hash of (seed, coords...) -> uint64 array (broadcasting)."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed))
        for cdim in coords:
            h = splitmix64(h ^ (np.asarray(cdim, dtype=np.uint64) * _GOLDEN))
        return h


def uniform01(seed, *coords):
    return (hash_u64(seed, *coords) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def clustered_signatures(cell_count, lsh_count, cluster_count=64, flip=0.15, seed=12345):
    """Scan-only input: signature = cluster centre with every bit flipped with probability `flip`.
    Returns uint64 [cell_count, words], bit i of a cell in word i>>6 at position 63-(i&63)."""
    words = (lsh_count - 1) // 64 + 1
    cells = np.arange(cell_count, dtype=np.uint64)
    cluster = hash_u64(seed, 1, cells) % np.uint64(cluster_count)
    out = np.zeros((cell_count, words), dtype=np.uint64)
    chunk = max(1, (1 << 22) // max(1, lsh_count))
    bits = np.arange(lsh_count, dtype=np.uint64)
    for begin in range(0, cell_count, chunk):
        end = min(cell_count, begin + chunk)
        cl = cluster[begin:end, None]
        centre = (hash_u64(seed, 2, cl, bits[None, :]) & np.uint64(1)).astype(np.uint8)
        flips = (uniform01(seed, 3, cells[begin:end, None], bits[None, :]) < flip).astype(np.uint8)
        b = centre ^ flips
        pad = words * 64 - lsh_count
        if pad:
            b = np.concatenate([b, np.zeros((end - begin, pad), dtype=np.uint8)], axis=1)
        packed = np.packbits(b, axis=1, bitorder="big")          # first bit most significant
        out[begin:end] = packed.reshape(end - begin, words, 8).view(">u8").reshape(end - begin, words)
    return out


def random_signatures(cell_count, lsh_count, seed=7):
    words = (lsh_count - 1) // 64 + 1
    idx = np.arange(cell_count * words, dtype=np.uint64).reshape(cell_count, words)
    sig = hash_u64(seed, 9, idx)
    pad = words * 64 - lsh_count
    if pad:
        sig[:, -1] &= ~np.uint64((1 << pad) - 1)
    return sig


def expression_matrix(cell_count, gene_count, density=0.01, cluster_count=64, seed=12345):
    """Clustered sparse expression matrix as CSR (toc uint64, genes uint32 ascending per cell, counts float32).
    Cell c belongs to cluster hash(c) % cluster_count; 70% of its genes come from the cluster's pool
    (2% of all genes), 30% are uniform; count = 1 + floor(-8 ln u)."""
    pool_size = max(4, gene_count // 50)
    toc = np.zeros(cell_count + 1, dtype=np.uint64)
    genes_all = []
    counts_all = []
    cells = np.arange(cell_count, dtype=np.uint64)
    cluster = hash_u64(seed, 11, cells) % np.uint64(cluster_count)
    nnz_target = np.maximum(1, np.rint(density * gene_count * (0.5 + uniform01(seed, 12, cells)))).astype(np.int64)
    max_n = int(nnz_target.max())
    j = np.arange(max_n, dtype=np.uint64)
    chunk = max(1, (1 << 21) // max_n)
    for begin in range(0, cell_count, chunk):
        end = min(cell_count, begin + chunk)
        cc = cells[begin:end, None]
        from_pool = uniform01(seed, 13, cc, j[None, :]) < 0.7
        pool_slot = hash_u64(seed, 14, cc, j[None, :]) % np.uint64(pool_size)
        pool_gene = hash_u64(seed, 15, cluster[begin:end, None], pool_slot) % np.uint64(gene_count)
        free_gene = hash_u64(seed, 16, cc, j[None, :]) % np.uint64(gene_count)
        g = np.where(from_pool, pool_gene, free_gene).astype(np.int64)
        valid = j[None, :].astype(np.int64) < nnz_target[begin:end, None]
        u = np.maximum(uniform01(seed, 17, cc, j[None, :]), 1e-300)
        cnt = (1.0 + np.floor(-8.0 * np.log(u))).astype(np.float32)
        for r in range(end - begin):
            gr = g[r][valid[r]]
            cr = cnt[r][valid[r]]
            ug, first = np.unique(gr, return_index=True)       # ascending, de-duplicated
            genes_all.append(ug.astype(np.uint32))
            counts_all.append(cr[first])
            toc[begin + r + 1] = toc[begin + r] + np.uint64(len(ug))
    genes = np.concatenate(genes_all) if genes_all else np.zeros(0, np.uint32)
    counts = np.concatenate(counts_all) if counts_all else np.zeros(0, np.float32)
    return toc, genes, counts

"""Synthetic sparse expression matrices generated directly in HBM with torch (bench / smoke tooling).

Shape follows SURVEY.md 8(d): 64 clusters, each with a gene pool of 2% of the genes; a cell draws about
density*geneCount*(0.5+u) genes, 70% from its cluster's pool and 30% uniformly; ids are de-duplicated and
ascending within a cell; count = 1 + floor(-8 ln u).  Returned as the CSR the C ABI takes: toc (int64,
relative to the first cell of the range) and data (int64 view of em2_count {gene:u32, count:f32})."""
import numpy as np


def expression_shard(cell_begin, cell_end, gene_count, density=0.01, cluster_count=64, seed=12345,
                     device="cuda", chunk_cells=32768):
    import torch
    pool_size = max(4, gene_count // 50)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    pools = torch.randint(0, gene_count, (cluster_count, pool_size), generator=gen, device=device)
    max_n = int(density * gene_count * 1.5) + 2
    toc_parts = [torch.zeros(1, dtype=torch.int64, device=device)]
    data_parts = []
    offset = 0
    for begin in range(cell_begin, cell_end, chunk_cells):
        end = min(cell_end, begin + chunk_cells)
        n = end - begin
        gen.manual_seed(seed * 1000003 + begin)          # shard boundaries do not change a cell's content
        cluster = torch.randint(0, cluster_count, (n,), generator=gen, device=device)
        target = torch.clamp(torch.round(density * gene_count * (0.5 + torch.rand(n, generator=gen, device=device))),
                             min=1).to(torch.int64)
        from_pool = torch.rand((n, max_n), generator=gen, device=device) < 0.7
        slot = torch.randint(0, pool_size, (n, max_n), generator=gen, device=device)
        pool_gene = pools[cluster.unsqueeze(1), slot]
        free_gene = torch.randint(0, gene_count, (n, max_n), generator=gen, device=device)
        gene = torch.where(from_pool, pool_gene, free_gene)
        valid = torch.arange(max_n, device=device).unsqueeze(0) < target.unsqueeze(1)
        gene = torch.where(valid, gene, torch.full_like(gene, gene_count))
        gene, _ = torch.sort(gene, dim=1)
        keep = gene < gene_count
        keep[:, 1:] &= gene[:, 1:] != gene[:, :-1]
        u = torch.rand((n, max_n), generator=gen, device=device).clamp_min(1e-30)
        count = (1.0 + torch.floor(-8.0 * torch.log(u))).to(torch.float32)
        per_cell = keep.sum(dim=1)
        toc_parts.append(offset + torch.cumsum(per_cell, 0))
        offset += int(per_cell.sum().item())
        g = gene[keep].to(torch.int64)
        c = count[keep].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        data_parts.append(g | (c << 32))
    toc = torch.cat(toc_parts)
    data = torch.cat(data_parts) if data_parts else torch.zeros(0, dtype=torch.int64, device=device)
    return toc, data


def csr_to_host(toc, data):
    """(toc uint64, genes uint32, counts float32) numpy copies, e.g. for the CPU oracle."""
    d = data.cpu().numpy().view(np.uint64)
    genes = (d & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    counts = (d >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return toc.cpu().numpy().astype(np.uint64), genes, counts

"""BASELINE configs[0]: the reference's ToyTest1 expression matrix (tests/golden/ToyTest1_ExpressionMatrix.csv is the
data file /root/reference/tests/ToyTest1/ExpressionMatrix.csv), read the way ExpressionMatrix::addCells stores it:
genes = CSV rows in file order (ids 0..), cells = CSV columns, zero counts dropped (src/ExpressionMatrix.cpp:254-256),
each cell's counts sorted by gene id (:266-267)."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ToyTest1_ExpressionMatrix.csv")


def load():
    with open(PATH) as f:
        rows = [line.strip().split(",") for line in f if line.strip()]
    cell_names = rows[0][1:]
    gene_count = len(rows) - 1
    toc, genes, counts = [0], [], []
    for c in range(len(cell_names)):
        for g in range(gene_count):
            value = float(rows[1 + g][1 + c])
            if value != 0.:
                genes.append(g)
                counts.append(value)
        toc.append(len(genes))
    return gene_count, np.array(toc, dtype=np.uint64), np.array(genes, dtype=np.uint32), np.array(counts, dtype=np.float32)

#!/usr/bin/env python3
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_binding
from expressionmatrix2_amd import capi
import synth
oracle = oracle_binding.load_oracle()
case = {'n': 1500, 'L': 1024, 'k': 5, 'thr': 0.5, 'clusters': 1, 'flip': 0.1, 'sig_seed': 285543329}
knobs = {'EM2_SCAN_MODE': 'triangle', 'EM2_MIN_SEGMENT_COLUMNS': '257', 'EM2_LOG_CAPACITY': '256', 'EM2_FULL_ROW_CELLS': '64', 'EM2_BLOCKS_PER_CU': '2', 'EM2_SCAN_MATRIX': '1'}
os.environ.update(knobs)
sig = synth.clustered_signatures(case['n'], case['L'], cluster_count=case['clusters'], flip=case['flip'], seed=case['sig_seed'])
cell, sim, used = oracle.find_similar_pairs4(sig, case['L'], case['k'], case['thr'])
bits = np.unpackbits(sig.view(np.uint8), axis=1)
def mism(a, b): return int((bits[a] != bits[b]).sum())
for convoy in sys.argv[1:] or ["0", "5"]:
    os.environ["EM2_MATRIX_CONVOY"] = convoy
    pairs, gused = capi.find_similar_pairs4(sig, case['L'], case['k'], case['thr'])
    bad = np.nonzero((gused != used) | (pairs["cell"] != cell).any(axis=1))[0]
    print("convoy", convoy, "differing rows", len(bad), bad[:40].tolist())
    for r in bad[:6]:
        print("  row", r, "got", [(int(c), round(float(s), 4), mism(r, int(c))) for c, s in zip(pairs["cell"][r], pairs["similarity"][r])])
        print("       exp", [(int(c), round(float(s), 4), mism(r, int(c))) for c, s in zip(cell[r], sim[r])])
    # which row blocks differ
    if len(bad):
        print("  blocks of 64 with differences:", sorted(set((bad // 64).tolist())))

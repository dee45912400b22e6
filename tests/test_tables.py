"""em2_tables (integer form of the reference's floating-point acceptance rules) against a direct evaluation of
the rules of src/ExpressionMatrixLsh.cpp:207,244-257 in numpy float64/float32."""
import numpy as np
import pytest


@pytest.mark.parametrize("L", [1, 2, 64, 100, 128, 1000, 1024, 2048, 4096])
@pytest.mark.parametrize("thr", [0.2, 0.0, -0.5, 0.5, 0.999, 1.0, -1.0, 0.19999999, float(np.float32(0.2))])
def test_tables_match_direct_rules(hostchecks, oracle, L, thr):
    t = hostchecks.tables(L, thr)
    sim = oracle.similarity_table(L)
    assert np.array_equal(t["similarity"].view(np.uint64), sim.view(np.uint64))
    simf = sim.astype(np.float32)
    # keys: rank of the float value
    assert np.array_equal(t["key_similarity"][t["key_of_mismatch"]].view(np.uint32), simf.view(np.uint32))
    assert np.all(np.diff(t["key_similarity"]) < 0)
    m = np.arange(L + 1)
    passes_global = sim > thr
    assert t["m_global"] == (m[passes_global].max() if passes_global.any() else -1)
    cell_thr = np.float32(thr)
    ok = passes_global & (sim > np.float64(cell_thr))
    assert t["m_max_initial"] == (m[ok].max() if ok.any() else -1)
    # and the accepted set is exactly m <= m_max_initial (monotone table)
    assert np.array_equal(ok, m <= t["m_max_initial"])
    for q, value in enumerate(t["key_similarity"]):
        ok = passes_global & (sim > np.float64(value))
        expect = m[ok].max() if ok.any() else -1
        assert t["accept_max_by_key"][q] == expect
        assert np.array_equal(ok, m <= expect)


def test_equal_mismatch_acceptance_quirk_exists(hostchecks):
    """The double-vs-float comparison accepts a candidate whose mismatch count EQUALS the cut-off's whenever
    float(cos) rounded down (SURVEY.md 'double-vs-float comparison quirk'); make sure the tables carry it."""
    t = hostchecks.tables(1024, 0.2)
    same = [q for q in range(len(t["key_similarity"]))
            if t["accept_max_by_key"][q] >= 0 and t["key_of_mismatch"][t["accept_max_by_key"][q]] == q]
    assert len(same) > 50

"""em2_select.h (the product's restatement of libstdc++ introselect, used on the device) against the real
std::nth_element / std::__introselect on the host, including the heap-select fallback that only a forced depth
limit reaches, and against the reference's keepBest."""
import numpy as np
import pytest

import synth


def entries(n, distinct, seed):
    idx = np.arange(n, dtype=np.uint64)
    cell = (synth.hash_u64(seed, 21, idx) % np.uint64(1 << 24)).astype(np.uint32)
    key = (synth.hash_u64(seed, 22, idx) % np.uint64(distinct)).astype(np.uint32)
    return cell, key


@pytest.mark.parametrize("n,nth,distinct", [(200, 100, 3), (200, 100, 17), (200, 100, 1000), (199, 100, 9),
                                             (7, 3, 2), (4, 2, 4), (3, 1, 2), (2, 1, 2), (1, 0, 1), (50, 0, 5),
                                             (50, 49, 5), (1000, 100, 30), (8192, 4096, 50), (101, 100, 4)])
def test_matches_std_nth_element(hostchecks, n, nth, distinct):
    for seed in range(25):
        cell, key = entries(n, distinct, seed)
        a = hostchecks.nth_element(cell, key, nth)
        b = hostchecks.std_introselect(cell, key, nth)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("n,nth,distinct", [(200, 100, 5), (200, 100, 300), (33, 7, 3), (64, 63, 8), (10, 0, 2)])
def test_heap_select_fallback_matches_std(hostchecks, depth, n, nth, distinct):
    for seed in range(15):
        cell, key = entries(n, distinct, 100 + seed)
        a = hostchecks.nth_element(cell, key, nth, depth)
        b = hostchecks.std_introselect(cell, key, nth, depth)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_sorted_and_reverse_inputs(hostchecks):
    for n in (200, 201, 64):
        cell = np.arange(n, dtype=np.uint32)
        for key in (np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32)[::-1].copy(),
                    np.zeros(n, dtype=np.uint32), (np.arange(n, dtype=np.uint32) // 7)):
            a = hostchecks.nth_element(cell, key, n // 2)
            b = hostchecks.std_introselect(cell, key, n // 2)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_key_form_equals_reference_float_form(hostchecks, reflib):
    """(cell,key) with key = rank of the float similarity must move exactly like the reference's
    pair<CellId,float> under OrderPairsBySecondGreater."""
    table = np.cos(np.arange(1025) * np.pi / 1024.0).astype(np.float32)
    for seed in range(30):
        cell, m = entries(200, 40, 500 + seed)
        m = m + 300
        sim = table[m]
        rc, rs = reflib.keep_best(cell, sim, 100)
        ec, ek = hostchecks.nth_element(cell, m, 100)
        assert np.array_equal(ec[:100], rc)
        assert np.array_equal(table[ek[:100]].view(np.uint32), rs.view(np.uint32))


@pytest.mark.parametrize("n,nth,distinct", [(200, 100, 3), (200, 100, 17), (200, 100, 1000), (199, 100, 9),
                                             (7, 3, 2), (4, 2, 4), (5, 2, 1), (50, 0, 5), (50, 49, 5),
                                             (1000, 100, 30), (8192, 4096, 50), (101, 100, 4), (130, 64, 2),
                                             (64, 32, 64), (65, 1, 3), (4000, 2000, 7)])
def test_wave_parallel_model_matches_std(hostchecks, n, nth, distinct):
    """The wave-parallel partition formulation (model of csrc/em2_select_wave.h) == std::nth_element."""
    for seed in range(25):
        cell, key = entries(n, distinct, 7000 + seed)
        a = hostchecks.wave_model_nth_element(cell, key, nth)
        b = hostchecks.std_introselect(cell, key, nth)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("depth", [0, 1, 3])
def test_wave_parallel_model_depth_limit(hostchecks, depth):
    for seed in range(20):
        cell, key = entries(300, 6, 9000 + seed)
        a = hostchecks.wave_model_nth_element(cell, key, 150, depth)
        b = hostchecks.std_introselect(cell, key, 150, depth)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_wave_parallel_model_structured_inputs(hostchecks):
    for n in (200, 201, 64, 129):
        cell = np.arange(n, dtype=np.uint32)
        for key in (np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32)[::-1].copy(),
                    np.zeros(n, dtype=np.uint32), (np.arange(n, dtype=np.uint32) // 7),
                    (np.arange(n, dtype=np.uint32) % 2), (np.arange(n, dtype=np.uint32) % 3)):
            for nth in (0, 1, n // 2, n - 1):
                a = hostchecks.wave_model_nth_element(cell, key, nth)
                b = hostchecks.std_introselect(cell, key, nth)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])

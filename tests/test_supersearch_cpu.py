"""CPU tests of the shared evolutionary search (adt_amd/supersearch.py): block-choice arithmetic against values recorded from the
reference supernets (tests/golden/super*.npz hold the reference's shared_idx / shared_weights for their candidates), and the
population logic over batched evaluation with a synthetic scoring function."""
import os
import random

import numpy as np

from adt_amd.supersearch import EvolutionSearch, cand_to_block, get_shared, get_weight

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_block_choice_matches_reference_fixtures():
    for name in ("super_c3", "super_l2", "superbert_c3", "superbert_l2", "superstosa_c3", "superstosa_l2"):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        block, rec_w, ind_w = cand_to_block(g["rec_choice"], g["ind_choice"], [float(x) for x in g["cand"]])
        sh = get_shared(g["rec_choice"], g["ind_choice"], block)
        assert [list(s[0]) for s in sh] == g["shared_idx"].tolist(), name
        assert np.allclose([list(s[1]) for s in sh], g["shared_weights"], rtol=0, atol=1e-12), name
        assert all(abs(sum(s[1]) - 1.0) < 1e-12 for s in sh)


def test_get_weight_is_piecewise_linear():
    ch = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
    assert get_weight(ch, 0.0) == 0 and abs(get_weight(ch, 0.2) - 0.0001) < 1e-15 and abs(get_weight(ch, 0.5) - 0.00075) < 1e-12
    assert abs(get_weight(ch, 0.1) - 0.00005) < 1e-15


def _run(seed, chunk=10):
    random.seed(seed)
    np.random.seed(seed)
    calls = []

    def evaluate(cands):          # score = closeness to 0.6 in every coordinate: a smooth optimum the search must move towards
        calls.append(len(cands))
        return [{"auc": 1.0 - float(np.mean((np.asarray(c) - 0.6) ** 2))} for c in cands]
    s = EvolutionSearch(2, evaluate, "auc", select_num=6, population_num=12, m_prob=0.3, crossover_num=3, mutation_num=3, scale_factor=0.5,
                        chunk=chunk)
    top = s.run(6)
    return s, top, calls


def test_search_population_and_batching():
    s, top, calls = _run(5)
    assert len(top) == 6 and len(s.candidates) == 12
    scores = [s.vis_dict[str(c)]["auc"] for c in top]
    assert scores == sorted(scores, reverse=True)
    assert all(len(c) == 4 and all(0.0 < x < 1.0 for x in c) for c in top)
    # every scored candidate was scored exactly once, in batched passes (several candidates per evaluation call)
    assert sum(calls) == s.evaluated == sum(1 for v in s.vis_dict.values() if "auc" in v)
    assert s.batches == len(calls) and max(calls) > 1 and len(calls) < s.evaluated
    # the search improves on the initial random population
    first = max(s.vis_dict[str(c)]["auc"] for c in s.memory[0])
    assert scores[0] >= first


def test_search_is_deterministic_given_the_seeds():
    a = _run(9)[1]
    b = _run(9)[1]
    assert a == b
    assert _run(9, chunk=1)[1] != [] and len(_run(9, chunk=1)[2]) >= len(_run(9)[2])      # chunk = 1 is the reference's one-at-a-time granularity

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


def test_propose_visits_exactly_what_the_reference_generator_consumes():
    """`propose` against a line-by-line restatement of the reference's lazy loop (sasrec/evolution.py:162-170 stack_random_cand,
    :139-160 check_cand, :192-206 get_random): same destination list, same visited set (proposals of the last chunk that the walk never
    reached stay unvisited and unscored), same number of random calls -- with duplicates inside a chunk and already-visited proposals."""
    pool = [[round(0.1 * (i % 7) + 0.01, 3), round(0.05 * (i % 5) + 0.02, 3)] for i in range(40)]      # plenty of duplicates

    def make(seed):
        r = random.Random(seed)
        return lambda: list(pool[r.randrange(len(pool))])

    def score(c):
        return 1.0 - (c[0] - 0.3) ** 2 - (c[1] - 0.1) ** 2

    for seed, quota, max_iter in ((1, 7, 400), (2, 3, 400), (3, 12, 9), (4, 5, 400)):
        # reference semantics, one candidate at a time
        vis, dest, calls, f = {}, [], 0, make(seed)
        vis[str(pool[0])] = {"visited": True, "auc": score(pool[0])}          # something visited before the call
        it = max_iter

        def gen():
            nonlocal calls
            while True:
                cands = [f() for _ in range(10)]
                calls += 10
                for c in cands:
                    vis.setdefault(str(c), {})
                for c in cands:
                    yield c
        g = gen()
        while len(dest) < quota and it > 0:
            it -= 1
            c = next(g)
            info = vis[str(c)]
            if "visited" in info:
                continue
            info["visited"] = True
            info["auc"] = score(c)
            dest.append(c)
        # batched implementation
        ncalls = [0]
        f2 = make(seed)

        def rf():
            ncalls[0] += 1
            return f2()
        s = EvolutionSearch(1, lambda cs: [{"auc": score(c)} for c in cs], "auc", select_num=2, population_num=quota, m_prob=0.3, crossover_num=1,
                            mutation_num=1, scale_factor=0.5, chunk=10)
        s.vis_dict[str(pool[0])] = {"visited": True, "auc": score(pool[0])}
        got = s.propose(rf, [], quota, max_iter)
        assert got == dest, (seed, got, dest)
        assert ncalls[0] == calls
        assert {k for k, v in s.vis_dict.items() if "visited" in v} == {k for k, v in vis.items() if "visited" in v}
        assert set(s.vis_dict) == set(vis)
        assert all(abs(s.vis_dict[k]["auc"] - vis[k]["auc"]) < 1e-15 for k, v in vis.items() if "auc" in v)

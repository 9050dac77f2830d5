"""Machinery shared by the three supernets (SASRec-ADT, BERT4Rec-ADT, STOSA-ADT) and their evolutionary lambda searches.

The reference carries three copies of the same search (sasrec/evolution.py:60-360, bert4rec/evolution.py:36-347,
stosa/searcher.py:23-279) that score ONE candidate at a time: `check_cand` runs a full validation pass of the supernet under
that candidate's block choice.  Here the unit of work is a BATCH of candidates:

  * `candidate_features` runs the supernet's encoder stack for P candidates on one validation batch at once.  A candidate
    only selects, per depth, four of the rec_size * ind_size candidate layers and four mixing weights, so
      - at depth 0 every candidate reads the same embedded input: each distinct layer (for the chained STOSA form: each distinct
        chain prefix) is evaluated ONCE and shared by all candidates that select it (there are only (rec_size-1)*(ind_size-1)
        bins of four layers, so P = 100 candidates need at most 36 depth-0 layer evaluations instead of 400);
      - at depth >= 1 the inputs differ per candidate: all candidates that select layer j are stacked along the batch axis and
        layer j runs once on the stack (one launch sequence per distinct layer instead of one per (candidate, layer)).
  * `EvolutionSearch` keeps the reference's population semantics (chunks of ten proposals, visited set, top-k, differential
    mutation, uniform crossover, random refill, the same numpy / random call order per proposal) but hands every chunk's
    unvisited proposals to ONE batched evaluation.

Host Python only: the layers themselves run in libadt_hip.so through the models' own layer functions.
"""
import json
import os
import random

import numpy as np
import torch

from . import ops


# ---- candidate -> block choice (sasrec/base_super_modules.py:15-57; identical copies in bert4rec/model and stosa/super_modules.py) ----
def get_position(weight, choice):
    """The interval of `choice` that holds `weight` and the two interpolation weights."""
    i1 = int(np.where(choice > weight)[0][0])
    i0 = i1 - 1
    p0 = (weight - choice[i0]) / (choice[i1] - choice[i0])
    return i0, i1, p0, 1 - p0


def get_shared(rec_choice, ind_choice, block_cand):
    """Per depth: the four candidate-layer indices (both pairs strided by rec_size, as the reference computes them) and their
    bilinear weights (p1 p3, p0 p3, p1 p2, p0 p2)."""
    rec_choice, ind_choice = np.asarray(rec_choice, np.float64), np.asarray(ind_choice, np.float64)
    rs = len(rec_choice)
    out = []
    for i in range(len(block_cand) // 2):
        i0, i1, p0, p1 = get_position(block_cand[2 * i], rec_choice)
        i2, i3, p2, p3 = get_position(block_cand[2 * i + 1], ind_choice)
        out.append(((i0 * rs + i2, i1 * rs + i2, i0 * rs + i3, i1 * rs + i3), (p1 * p3, p0 * p3, p1 * p2, p0 * p2)))
    return out


def get_weight(choices, prob):
    """SearcherEvolution._get_weight: piecewise-linear map of a probability in [0, 1] onto the loss-weight grid."""
    split = 1 / (len(choices) - 1)
    idx = 0
    while prob > split:
        idx += 1
        prob -= split
    rd = prob / split
    return choices[idx] * (1 - rd) + choices[idx + 1] * rd


def cand_to_block(rec_choice, ind_choice, cand):
    """SearcherEvolution._set_choice: probabilities -> (block choice for the model, rec loss weights, ind loss weights)."""
    block, rec_w, ind_w = [], [], []
    for i in range(0, len(cand), 2):
        rw, iw = get_weight(rec_choice, cand[i]), get_weight(ind_choice, cand[i + 1])
        rec_w.append(rw)
        ind_w.append(iw)
        block += [rw, iw]
    return np.array(block), rec_w, ind_w


# ---- batched candidate evaluation ---------------------------------------------------------------------------------------------
def _stack(parts):
    """parts: list of tuples of (rows, d) tensors -> one tuple of stacked tensors."""
    if len(parts) == 1:
        return parts[0]
    return tuple(torch.cat([p[k] for p in parts], 0) for k in range(len(parts[0])))


def _mix(outs, ws):
    """sum_k w_k * outs[k] for tuples of tensors (k_axpy, the same kernel the training-time mixing uses)."""
    res = []
    for t in range(len(outs[0])):
        acc = torch.empty_like(outs[0][t])
        for k, w in enumerate(ws):
            ops.axpy(acc, outs[k][t], float(w), k > 0)
        res.append(acc)
    return tuple(res)


def candidate_features(run_layer, x0, shared_list, num_layers, chain=False, stats=None):
    """Encoder-stack output of every candidate on one batch.

    run_layer(depth, idx, x, n) -> y : candidate layer `idx` of `depth` on a tuple `x` of tensors that holds n stacked copies of
    the batch (rows n * B * L), returning a tuple of the same structure.  x0: the embedded input (tuple).  shared_list[p][depth] =
    (four layer indices, four weights) of candidate p.  chain: the four selected layers are applied one after the other and the
    four intermediate results are mixed (stosa/super_modules.py:75-95); otherwise all four read the layer input
    (sasrec/super_modules.py:35-50, bert4rec/model/modules.py:243-259).  Returns [tuple per candidate]."""
    P = len(shared_list)
    rows = x0[0].shape[0]
    cur = [x0] * P
    for depth in range(num_layers):
        outs = [[None] * 4 for _ in range(P)]
        for k in range(4):
            groups = {}      # (layer idx, identity of the input) -> candidates
            for p in range(P):
                idxs = shared_list[p][depth][0]
                if chain and k > 0:
                    ikey = tuple(idxs[:k]) if depth == 0 else p
                else:
                    ikey = 0 if depth == 0 else p
                groups.setdefault((idxs[k], ikey), []).append(p)
            by_idx = {}
            for (idx, _), ps in groups.items():
                by_idx.setdefault(idx, []).append(ps)
            for idx, plist in by_idx.items():
                xs = [outs[ps[0]][k - 1] if (chain and k > 0) else cur[ps[0]] for ps in plist]
                y = run_layer(depth, idx, _stack(xs), len(xs))
                if stats is not None:
                    stats["layer_calls"] = stats.get("layer_calls", 0) + 1
                    stats["layer_copies"] = stats.get("layer_copies", 0) + len(xs)
                for j, ps in enumerate(plist):
                    yj = tuple(t[j * rows:(j + 1) * rows] for t in y) if len(plist) > 1 else y
                    for p in ps:
                        outs[p][k] = yj
        memo = {}
        nxt = []
        for p in range(P):
            key = shared_list[p][depth] if depth == 0 else None      # same bin and same weights on the shared input: same mix
            if key is not None and key in memo:
                nxt.append(memo[key])
                continue
            m = _mix(outs[p], shared_list[p][depth][1])
            if key is not None:
                memo[key] = m
            nxt.append(m)
        cur = nxt
    return cur


# ---- the search ---------------------------------------------------------------------------------------------------------------
def _np_choice(x):
    """The reference's module-level `choice` (np.random.randint based; sasrec/evolution.py:27-28)."""
    return x[np.random.randint(len(x))]


class EvolutionSearch:
    """Population bookkeeping of the evolutionary lambda search over batched candidate evaluation.

    evaluate(cands) -> list of metric dicts, one per candidate, each with `score_key` (validation AUC for SASRec / BERT4Rec,
    MRR for STOSA) -- ONE batched pass for the whole list.  Proposals are drawn in chunks of `chunk` (ten, the granularity of the
    reference's stack_random_cand generator), duplicates of already visited candidates are dropped, the rest of the chunk is
    scored together and accepted in proposal order until the quota is met; `max_iter` bounds the proposals drawn, as in the
    reference's loops."""

    def __init__(self, num_layers, evaluate, score_key, select_num, population_num, m_prob, crossover_num, mutation_num, scale_factor,
                 chunk=10):
        self.num_layers, self.evaluate, self.score_key = num_layers, evaluate, score_key
        self.select_num, self.population_num, self.m_prob = select_num, population_num, m_prob
        self.crossover_num, self.mutation_num, self.scale_factor, self.chunk = crossover_num, mutation_num, scale_factor, chunk
        self.vis_dict, self.candidates, self.top, self.memory, self.epoch = {}, [], [], [], 0
        self.evaluated, self.batches = 0, 0

    # proposal generators (the reference's random_func closures)
    def sample_random(self):
        return [random.random() for _ in range(2 * self.num_layers)]

    def _crossover_one(self):
        c1, c2 = _np_choice(self.top), _np_choice(self.top)
        return [_np_choice([i, j]) for i, j in zip(c1, c2)]

    def _mutation_one(self):
        cand = list(_np_choice(self.top))
        for i in range(self.num_layers * 2):
            if np.random.random_sample() < self.m_prob:
                c2, c3 = list(_np_choice(self.top)), list(_np_choice(self.top))
                cand[i] = min(1 - 1e-10, max(1e-10, cand[i] + self.scale_factor * (c2[i] - c3[i])))
        return cand

    def _score_batch(self, fresh):
        """Scores candidates (already marked visited) in one batched supernet pass."""
        if fresh:
            for c, metrics in zip(fresh, self.evaluate(fresh)):
                self.vis_dict[str(c)].update(metrics)
            self.evaluated += len(fresh)
            self.batches += 1
        return fresh

    def propose(self, random_func, dest, quota, max_iter):
        """Fill `dest` up to `quota` with fresh scored candidates from random_func -- the reference's `stack_random_cand` + `check_cand` loop
        (sasrec/evolution.py:162-170, 192-206): proposals are drawn ten at a time (the same random call order), walked in order, one
        `max_iter` each, and a proposal is visited (and scored) only if the walk reaches it before the quota is full -- the rest of the last
        chunk stays unvisited, exactly as the reference's lazy generator leaves it.  What differs is only WHEN the scores are computed: the
        proposals the walk consumes from a chunk are scored together in one batched pass (they do not influence each other's scores)."""
        while len(dest) < quota and max_iter > 0:
            chunk = [random_func() for _ in range(self.chunk)]
            for c in chunk:
                self.vis_dict.setdefault(str(c), {})
            take, need = [], quota - len(dest)
            for c in chunk:
                if max_iter <= 0 or len(take) >= need:
                    break
                max_iter -= 1
                info = self.vis_dict[str(c)]
                if "visited" in info:
                    continue
                info["visited"] = True
                take.append(c)
            dest.extend(self._score_batch(take))
        return dest

    def get_random(self):
        return self.propose(self.sample_random, self.candidates, self.population_num, (self.population_num - len(self.candidates) + 1) * 50)

    def update_top_k(self):
        t = self.top + self.candidates
        t.sort(key=lambda c: self.vis_dict[str(c)][self.score_key], reverse=True)
        self.top = t[:self.select_num]

    def run(self, search_epochs, log=None):
        self.get_random()
        for _ in range(search_epochs):
            self.epoch += 1
            self.memory.append(list(self.candidates))
            self.update_top_k()
            mutation = self.propose(self._mutation_one, [], self.mutation_num, self.mutation_num * 10)
            crossover = self.propose(self._crossover_one, [], self.crossover_num, self.crossover_num * 10)
            self.candidates = mutation + crossover
            self.get_random()
            if log:
                best = self.vis_dict[str(self.top[0])][self.score_key] if self.top else float("nan")
                log("search epoch %d: best %s %.5f, %d candidates scored in %d batched passes" % (self.epoch, self.score_key, best, self.evaluated,
                                                                                             self.batches))
        return self.top

    def write(self, path, rec_choice, ind_choice):
        """One JSON object per surviving candidate (the reference writes the same records through `jsonlines`)."""
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            for cand in self.top:
                info = dict(self.vis_dict[str(cand)])
                _, rec_w, ind_w = cand_to_block(rec_choice, ind_choice, cand)
                info["cand"], info["rec"], info["ind"] = str(cand), str([float(x) for x in rec_w]), str([float(x) for x in ind_w])
                f.write(json.dumps(info) + "\n")
        return path


def result_name(out_dir, a):
    return os.path.join(out_dir, "res_%s_lr_%s_reg_%s_warm_%d_search_%d_layers_%d_select_%d_population_%d_cross_%d_mutation_%d.jsonl" % (
        a.dataset, a.lr, a.weight_decay, a.warmup_epochs, a.search_epochs, a.num_layers, a.select_num, a.population_num, a.crossover_num,
        a.mutation_num))

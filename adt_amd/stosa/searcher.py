"""SearcherEvolution for STOSA-ADT on the MI355X path -- the counterpart of the reference's stosa/searcher.py:23-279: warm up the
weight-sharing supernet with one random candidate per epoch (:235-240), then evolve a population of (reconstruction, independence)
weight candidates scored by the validation MRR of the full-sort ranking under each candidate's block choice (get_cand_MRR :123-129).

The supernet and its warm-up step run in libadt_hip.so (adt_amd/stosa/supernet.py); candidates are scored a chunk at a time -- one
batched pass of the validation set per chunk (adt_amd/supersearch.py), the full sort and the seen-item masking on the device.
"""
import os

import numpy as np
import torch

from .. import ops
from ..supersearch import EvolutionSearch, cand_to_block, get_shared, result_name
from .datasets import DisenDataset, get_user_seqs
from .supernet import DisenDistSASupernet, SuperStosaTrainer
from .trainer import get_full_sort_score


class SearcherEvolution:
    def __init__(self, args):
        self.args = args
        user_seq, max_item, valid_matrix, test_matrix, num_users = get_user_seqs(args.data_file)
        args.item_size, args.num_users, args.mask_id = max_item + 2, num_users, max_item + 1
        self.valid_matrix, self.test_matrix = valid_matrix, test_matrix
        self.train_ds = DisenDataset(args, user_seq, "train", seed=args.seed)
        self.valid_ds = DisenDataset(args, user_seq, "valid", args.eval_set, seed=args.seed + 1)
        self.test_ds = DisenDataset(args, user_seq, "test", args.eval_set, seed=args.seed + 2)
        # search space (stosa/searcher.py:54-55: both grids are the reconstruction grid)
        self.rec_choice = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
        self.ind_choice = [0, 0.0001, 0.0005, 0.001, 0.005, 0.01]
        torch.manual_seed(args.seed)
        self.model = DisenDistSASupernet(args, self.rec_choice, self.ind_choice)
        self.trainer = SuperStosaTrainer(self.model, lr=args.lr, betas=(args.adam_beta1, args.adam_beta2), weight_decay=args.weight_decay,
                                         seed=args.seed)
        self.search_state = EvolutionSearch(args.num_layers, self.evaluate_candidates, "MRR", args.select_num, args.population_num, args.m_prob,
                                            args.crossover_num, args.mutation_num, args.scale_factor)
        self.eval_stats = {}

    @property
    def vis_dict(self):
        return self.search_state.vis_dict

    def _seen_csr(self, matrix, users, copies):
        """CSR of the users' seen items, repeated for `copies` stacked candidates, on the device."""
        csr = matrix[users].tocsr()
        ip, ix = csr.indptr.astype(np.int64), csr.indices
        if ix.size == 0:
            return None, None
        n = len(users)
        ip_all = np.concatenate([ip[:-1] + k * ix.size for k in range(copies)] + [[copies * ix.size]])
        dev = self.model.dev
        assert ip_all.size == copies * n + 1
        return (torch.from_numpy(np.ascontiguousarray(ip_all, dtype=np.int32)).to(dev),
                torch.from_numpy(np.ascontiguousarray(np.tile(ix, copies), dtype=np.int32)).to(dev))

    def evaluate_candidates(self, cands, dataset=None, matrix=None, group=8, prefix="V"):
        """Full-sort scores (Trainer.get_full_sort_score, stosa/trainer.py:62-86) of the supernet under every candidate of `cands`:
        every validation batch is ranked for `group` candidates per pass (distances, seen-item masking and top-40 on the device)."""
        ds = self.valid_ds if dataset is None else dataset
        matrix = self.valid_matrix if matrix is None else matrix
        shared = [get_shared(self.rec_choice, self.ind_choice, cand_to_block(self.rec_choice, self.ind_choice, c)[0]) for c in cands]
        preds = [[] for _ in cands]
        answers = []
        for users, inp, dec, pos, neg, ans in ds.epoch_batches(self.args.eval_batch_size, shuffle=False):
            answers.append(np.asarray(ans))
            B = len(users)
            for g0 in range(0, len(cands), group):
                sl = shared[g0:g0 + group]
                dist = self.model.predict_full_candidates(inp, sl, stats=self.eval_stats)
                indptr, indices = self._seen_csr(matrix, users, len(sl))
                top = ops.topk_masked(dist, 40, indptr, indices).cpu().numpy().astype(np.int64)
                for k in range(len(sl)):
                    preds[g0 + k].append(top[k * B:(k + 1) * B])
        answers = np.concatenate(answers)
        out = []
        for pk in preds:
            s = get_full_sort_score(answers, np.concatenate(pk))
            out.append({prefix + "_NDCG": float(s[5]), prefix + "_HR": float(s[4]), prefix + "_MRR": float(s[-1]), "MRR": float(s[-1])})
        return out

    def _train_warmup(self):
        for epoch in range(self.args.warmup_epochs):
            self.trainer.set_choice(self.search_state.sample_random())
            for users, inp, dec, pos, neg, _ in self.train_ds.epoch_batches(self.args.batch_size):
                self.trainer.step(inp, dec, pos, neg)
            print("warmup epoch %d / %d loss %.4f" % (epoch + 1, self.args.warmup_epochs, float(self.trainer.loss())), flush=True)

    def search(self):
        self._train_warmup()
        os.makedirs("./checkpoint", exist_ok=True)
        torch.save(self.model.state_dict(), "./checkpoint/super.pth")
        self.search_state.run(self.args.search_epochs, log=lambda m: print(m, flush=True))
        return self.search_state.write(result_name(getattr(self.args, "out_dir", "res"), self.args), self.rec_choice, self.ind_choice)

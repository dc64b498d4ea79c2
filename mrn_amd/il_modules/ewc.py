"""EWC learner (reference il_modules/ewc.py:27-167): Fisher diagonal after every task, quadratic penalty afterwards.

Reference behaviour worth knowing: the Fisher dict is keyed by the DataParallel-prefixed names
("module....", :129) while compute_ewc iterates self.model.module.named_parameters() (:123), so no key ever matches
and the penalty is identically 0.  `reference_prefix_bug = True` (default) reproduces that literally (same two name
spaces); set it to False to apply the penalty the code intends (lamda * sum F (p - p*)^2 / 2) -- both run on the HIP
kernels over the flat buffers."""
import time

import torch

from .. import ops, parallel
from ..tools.utils import Averager
from .base import BaseLearner

lamda = 1000
fishermax = 0.0001
alpha = 0.5
num_iter = 5000


class EWC(BaseLearner):
    reference_prefix_bug = True
    fisher_iterations = num_iter

    def __init__(self, opt):
        super().__init__(opt)
        self.fisher = None
        self.mean = None

    def after_task(self):
        self.model = self.model.module
        self._known_classes = self._total_classes

    def _train(self, start_iter, taski, train_loader, valid_loader):
        if taski == 0:
            self._init_train(start_iter, taski, train_loader, valid_loader)
        else:
            self.load_task_data(train_loader, taski)
            self._update_representation(start_iter, taski, train_loader, valid_loader)
        if self.fisher is None:
            self.fisher = self.getFisherDiagonal(train_loader)
        else:
            # blend with the previous task's Fisher over the shared leading rows, paired BY POSITION in the two dicts (:48-55)
            new = self.getFisherDiagonal(train_loader)
            f_list = list(self.fisher.values())
            for i, n in enumerate(new):
                k = len(f_list[i])
                new[n][:k] = alpha * f_list[i] + (1 - alpha) * new[n][:k]
            self.fisher = new
        self.mean = {n: p.clone().detach() for n, p in self.model.named_parameters() if p.requires_grad}

    def _penalty_terms(self):
        """(fisher, parameter, rows, mean) for every parameter the penalty covers (ewc.py:120-126)"""
        if self.fisher is None:
            return []
        out = []
        for n, p in self.model.module.named_parameters():
            key = n if self.reference_prefix_bug else "module." + n      # reference: unprefixed name looked up in prefixed keys
            if key in self.fisher:
                k = len(self.mean[key])
                out.append((self.fisher[key][:k].contiguous(), p, k, self.mean[key]))
        return out

    def compute_ewc(self):
        total = torch.zeros((), device=self.device)
        for f, p, k, m in self._penalty_terms():
            total = total + ops.ewc_penalty(f.view(-1), p.detach()[:k].contiguous().view(-1), m.view(-1)).view(())
        return total

    def ewc_step(self, image, labels):
        """one iteration of ewc.py:73-104: loss = loss_clf + lamda * loss_ewc"""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        text = None if "CTC" in self.opt.Prediction else labels_index[:, :-1]
        preds = self.model(image, text, True)["predict"]
        loss_clf = self.criterion(preds, labels_index, labels_length)
        penalty = self.compute_ewc()

        def add_penalty_gradient():                           # d(lamda * penalty)/dp straight into the flat gradient
            for f, p, k, m in self._penalty_terms():
                if p.grad is not None:
                    ops.ewc_penalty_grad_(p.grad[:k].view(-1), f.view(-1), p.detach()[:k].view(-1), m.view(-1), lamda)
        self.backward_and_step(loss_clf, after_reduce=add_penalty_gradient)
        return loss_clf.detach() + lamda * penalty

    def _update_representation(self, start_iter, taski, train_loader, valid_loader):
        avg = Averager()
        start_time = time.time()
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image, labels = train_loader.get_batch()
            loss = self.ewc_step(image.to(self.device), labels)
            avg.add(loss.detach())
            self.end_iteration(iteration)
            if iteration % self.opt.val_interval == 0 or iteration == 1:
                self.val(valid_loader, self.opt, -1, start_time, iteration, avg, None, taski)
                avg.reset()

    def getFisherDiagonal(self, train_loader):
        """mean of squared gradients over `fisher_iterations` batches, clipped at fishermax (:128-167) -> {"module.<name>": tensor}
        (train mode: BatchNorm running statistics keep moving during these passes, as in the reference)"""
        flat = torch.zeros_like(self.optimizer.flat)
        self.model.train()
        for _ in range(self.fisher_iterations):
            image, labels = train_loader.get_batch()
            labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
            text = None if "CTC" in self.opt.Prediction else labels_index[:, :-1]
            preds = self.model(image.to(self.device), text, True)["predict"]
            loss = self.criterion(preds, labels_index, labels_length)
            self.optimizer.zero_grad()
            loss.backward()
            parallel.average_gradients(self.optimizer.grad)   # DataParallel: one gradient of the whole (gathered) batch
            ops.fisher_accumulate(flat, self.optimizer.grad)
        ops.fisher_finalize(flat, self.fisher_iterations, fishermax)
        index = {id(p): i for i, p in enumerate(self.optimizer.params)}
        out = {}
        for n, p in self.model.named_parameters():           # "module."-prefixed names, the reference's Fisher keys (:129)
            if p.requires_grad:
                out[n] = self.optimizer.view_of(flat, index[id(p)]).clone()
        return out

    get_fisher_diagonal = getFisherDiagonal

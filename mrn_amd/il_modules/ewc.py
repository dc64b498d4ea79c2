"""EWC learner (reference il_modules/ewc.py:27-167): Fisher diagonal after every task, quadratic penalty afterwards.

Reference behaviour worth knowing: the Fisher dict is keyed by the DataParallel-prefixed names
("module....", :129) while compute_ewc iterates self.model.module.named_parameters() (:123), so no key ever matches
and the penalty is identically 0.  `reference_prefix_bug = True` (default) reproduces that; set it to False to apply
the penalty the code intends (lamda * sum F (p - p*)^2 / 2) -- both run on the HIP kernels over the flat buffers."""
import time

import torch

from .. import ops
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

    def _named_trainable(self):
        return [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    def _train(self, start_iter, taski, train_loader, valid_loader):
        valid = valid_loader.create_dataset()
        if taski == 0:
            self._init_train(start_iter, taski, train_loader, valid)
        else:
            train_loader.get_dataset(taski, memory=self.opt.memory)
            self._update_representation(start_iter, taski, train_loader, valid)
        new = self.get_fisher_diagonal(train_loader)
        if self.fisher is not None:          # blend with the previous Fisher over the shared leading rows (:51-55)
            old = list(self.fisher.values())
            for i, n in enumerate(new):
                if i < len(old):
                    k = len(old[i])
                    new[n][:k] = alpha * old[i] + (1 - alpha) * new[n][:k]
        self.fisher = new
        self.mean = {n: p.detach().clone() for n, p in self._named_trainable()}

    def _penalty_terms(self):
        """(fisher, current[:len(mean)], mean) per parameter the penalty covers"""
        if self.reference_prefix_bug or self.fisher is None:
            return []                        # reference: "module."-prefixed Fisher keys never match (:123 vs :129)
        out = []
        for n, p in self._named_trainable():
            if n in self.fisher and n in self.mean:
                k = len(self.mean[n])
                out.append((self.fisher[n][:k].contiguous(), p, k, self.mean[n]))
        return out

    def compute_ewc(self):
        total = torch.zeros((), device=self.device)
        for f, p, k, m in self._penalty_terms():
            total = total + ops.ewc_penalty(f.view(-1), p.detach()[:k].contiguous().view(-1), m.view(-1)).view(())
        return total

    def ewc_step(self, image, labels):
        """one iteration of ewc.py:73-104: loss = loss_clf + lamda * loss_ewc"""
        from .. import parallel
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        text = None if "CTC" in self.opt.Prediction else labels_index[:, :-1]
        preds = self.model(image, text)["predict"]
        loss_clf = self.criterion(preds, labels_index, labels_length)
        penalty = self.compute_ewc()
        self.optimizer.zero_grad()
        loss_clf.backward()
        for f, p, k, m in self._penalty_terms():          # d(lamda * penalty)/dp added straight into the flat gradient
            g = p.grad[:k]
            ops.ewc_penalty_grad_(g.view(-1), f.view(-1), p.detach()[:k].view(-1), m.view(-1), lamda)
        parallel.average_gradients(self.optimizer.grad)
        lr = self.scheduler.lr_at(self.opt_step) if self.scheduler is not None else self.optimizer.lr
        self.optimizer.step(lr=lr, max_norm=self.opt.grad_clip)
        self.opt_step += 1
        return loss_clf + lamda * penalty

    def _update_representation(self, start_iter, taski, train_loader, valid_loader):
        avg = Averager()
        start_time, best = time.time(), -1
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image, labels = train_loader.get_batch()
            loss = self.ewc_step(image.to(self.device), labels)
            avg.add(loss.detach())
            if iteration % self.opt.val_interval == 0 or iteration == 1:
                best = self.val(valid_loader, self.opt, best, start_time, iteration, avg, None, taski)
                avg.reset()

    def get_fisher_diagonal(self, train_loader):
        """mean of squared gradients over `fisher_iterations` batches, clipped at fishermax (:128-167) -> {name: tensor}"""
        flat = torch.zeros_like(self.optimizer.flat)
        self.model.train()
        for _ in range(self.fisher_iterations):
            image, labels = train_loader.get_batch()
            labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
            text = None if "CTC" in self.opt.Prediction else labels_index[:, :-1]
            preds = self.model(image.to(self.device), text)["predict"]
            loss = self.criterion(preds, labels_index, labels_length)
            self.optimizer.zero_grad()
            loss.backward()
            ops.fisher_accumulate(flat, self.optimizer.grad)
        ops.fisher_finalize(flat, self.fisher_iterations, fishermax)
        out = {}
        base = self.optimizer.grad.data_ptr()
        for n, p in self._named_trainable():                # split the flat Fisher back into per-parameter tensors
            off = (p.grad.data_ptr() - base) // 4
            out[n] = flat[off:off + p.numel()].view(p.shape).clone()
        return out

"""MRN learner (reference il_modules/mrn.py:32-515): per task, step 0 trains the newest expert (loop A,
`_init_train(cross=False)`), the expert is frozen, step 1 trains the DM-Router over all frozen experts (loop B,
`_update_representation`, loss = 15 * loss_clf + CE(softmax weights, domain)).

`routing_step()` is one iteration of loop B -- the north-star hot path -- and is what bench.py times."""
import time

import torch

from .. import functional as Fn
from .. import parallel
from ..modules.model import MRNNet
from ..tools.utils import Averager, to_device
from .base import BaseLearner


class MRN(BaseLearner):
    checkpoint_has_step = True         # {lan}_{taski}_{step}_best_score.pth (mrn.py:414)

    def __init__(self, opt):
        super().__init__(opt)
        self.model = MRNNet(opt)

    def after_task(self):
        self.model = self.model.module
        self._known_classes = self._total_classes
        self._old_network = self.model.copy().freeze()

    def build_model(self):
        self.model.build_fc(self.opt.hidden_size, self._total_classes)
        self.model.build_prediction(self.opt, self._total_classes)
        self._reference_init()
        self._wrap(self.model)

    def freeze_experts(self, upto):
        for i in range(upto):
            for p in self.model.module.model[i].parameters():
                p.requires_grad = False

    def incremental_train(self, taski, character, train_loader, valid_loader):
        self.character = character
        self.converter = self.build_converter()
        if taski > 0:
            self.change_model()
        else:
            self.criterion = self.build_criterion()
            self.build_model()
        self.freeze_experts(taski)
        self.build_optimizer(self.count_param())
        self._train(0, taski, train_loader, valid_loader, step=0)
        if taski > 0:
            self._train(0, taski, train_loader, valid_loader, step=1)

    def memory_per_task(self, taski):
        """mrn.py:170-175: memories of 5000+ samples are per task, smaller ones are split over the previous tasks"""
        return self.opt.memory_num if self.opt.memory_num >= 5000 else int(self.opt.memory_num / taski)

    def build_rehearsal_memory(self, train_loader, taski):
        num_i = self.memory_per_task(taski)
        self.build_random_current_memory(num_i, taski, train_loader)
        if self.opt.memory_num < 5000:
            if len(self.memory_index) != 0 and len(self.memory_index) * len(self.memory_index[0]) > self.opt.memory_num:
                self.reduce_samplers(taski, taski_num=num_i)
        train_loader.get_dataset(taski, memory=self.opt.memory, index_list=self.memory_index)
        print("Is using rehearsal memory, has {} prev datasets, each has {}\n".format(len(self.memory_index), self.memory_index[0].size))

    def _train(self, start_iter, taski, train_loader, valid_loader, step=0):
        if self.opt.start_task > taski + step * 0.5:           # resume: load this (task, step)'s checkpoint (mrn.py:187-203)
            self.load_checkpoint(self.checkpoint_path(taski, step))
            if taski > 0 and step == 0:
                train_loader.get_dataset(taski, memory=None)
                for p in self.model.module.model[-1].parameters():     # what update_step1 leaves behind: newest expert frozen
                    p.requires_grad = False
                self.model.module.model[-1].eval()
            elif taski > 0 and step == 1:
                self.load_task_data(train_loader, taski)
            return
        print("Task {} start training for model ------{}------".format(taski, self.opt.exp_name))
        if taski == 0:
            self._init_train(start_iter, taski, train_loader, valid_loader.create_dataset(), cross=False)
        elif step == 0:
            train_loader.get_dataset(taski, memory=None)
            self.update_step1(start_iter, taski, train_loader, valid_loader.create_dataset())
        else:
            self.load_task_data(train_loader, taski)
            self._update_representation(start_iter, taski, train_loader, valid_loader.create_list_dataset())

    def _forward_train(self, image, text):
        return self.model(image, False, text)["logits"]          # cross=False: newest expert only (mrn.py:248,254)

    def _init_train(self, start_iter, taski, train_loader, valid_loader, cross=False):
        """mrn.py:225-279: loop A on the newest expert; validates at every val_interval and at the last iteration, "FF"""
        train_loss_avg = Averager()
        start_time = time.time()
        best_score = -1
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image_tensors, labels = train_loader.get_batch()
            loss = self.train_step(image_tensors.to(self.device), labels)
            train_loss_avg.add(loss.detach())
            self.end_iteration(iteration)
            if iteration % self.opt.val_interval == 0 or iteration == self.opt.num_iter:
                self.val(valid_loader, self.opt, best_score, start_time, iteration, train_loss_avg, None, taski, 0, "FF")
                train_loss_avg.reset()

    def update_step1(self, start_iter, taski, train_loader, valid_loader):
        self._init_train(start_iter, taski, train_loader, valid_loader, cross=False)
        for p in self.model.module.model[-1].parameters():
            p.requires_grad = False
        self.model.module.model[-1].eval()

    def test(self, AlignCollate_valid, valid_datas, best_scores, ned_scores, taski, val_choose="test"):
        # task 0 evaluates the single expert (checkpoint of step 0), later tasks the routed ensemble (step 1): mrn.py:450-466
        return super().test(AlignCollate_valid, valid_datas, best_scores, ned_scores, taski,
                            val_choose="FF" if taski == 0 else "TF", step=0 if taski == 0 else 1)

    # -- loop B ------------------------------------------------------------------------------------------
    def prepare_routing(self, total_steps=None):
        """optimiser of step 1: Adam over the router parameters, OneCycle(total = 2 * num_iter) (mrn.py:308-312)"""
        self.criterion = self.build_criterion()
        self.build_optimizer(self.count_param(), scale=1.0, total_steps=total_steps or self.opt.num_iter * 2, optimizer="adam",
                             schedule="super")            # build_custom_optimizer(optimizer="adam", schedule="super", the=2)

    def prefetch_experts(self, image, labels):
        """Issue the frozen experts' forward of a FUTURE loop-B batch (label encoding + MRNNet.experts_prefetch); pass the
        result to routing_step(..., prefetched=...).  The experts are frozen in loop B (mrn.py:285-286), so their forward
        for batch n+1 does not depend on the router update of batch n; only the launch order changes, not the results."""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        text = None if "CTC" in self.opt.Prediction else labels_index[:, :-1]
        return self.model.module.experts_prefetch(image, text, True), labels_index, labels_length

    def routing_step(self, image, labels, indexs, pi=15, prefetched=None):
        """one iteration of mrn.py:329-371"""
        if prefetched is not None:
            handle, labels_index, labels_length = prefetched
        else:
            handle = None
            labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        if "CTC" in self.opt.Prediction:
            output = self.model(image, True, experts=handle)
        else:
            output = self.model(image, True, labels_index[:, :-1], True, experts=handle)
        taski_loss = Fn.cross_entropy(output["index"], indexs, -100)     # CE on the already-softmaxed weights
        loss_clf = self.criterion(output["logits"], labels_index, labels_length)
        loss = pi * loss_clf + taski_loss
        self.optimizer_step(loss)
        return loss_clf, taski_loss

    def _update_representation(self, start_iter, taski, train_loader, valid_loader, pi=15):
        train_loss_avg, train_taski_loss_avg = Averager(), Averager()
        self.prepare_routing()
        start_time = time.time()
        best_score = -1
        n_iter = int(self.opt.num_iter // 2)
        def fetch():
            image_tensors, labels, indexs = train_loader.get_batch2()
            image = image_tensors.to(self.device)
            return image, labels, to_device(torch.LongTensor(indexs).squeeze()), self.prefetch_experts(image, labels)

        def validates(it):
            return it % max(self.opt.val_interval // 5, 1) == 0 or it == n_iter or it == 1

        # Software pipeline: batch n+1's frozen-expert forward is issued before batch n's router phase -- except across a
        # validation: the experts run train-mode BatchNorm here (mrn.py:107,401), so a look-ahead forward would advance their
        # running statistics before the evaluation that the reference runs first.
        nxt = fetch()
        for iteration in range(start_iter + 1, n_iter + 1):
            image, labels, indexs, pre = nxt
            nxt = None
            if iteration < n_iter and not validates(iteration):
                nxt = fetch()
            loss_clf, taski_loss = self.routing_step(image, labels, indexs, pi, prefetched=pre if pre[0] is not None else None)
            train_loss_avg.add(loss_clf.detach())
            train_taski_loss_avg.add(taski_loss.detach())
            if validates(iteration):
                self.val(valid_loader, self.opt, best_score, start_time, iteration, train_loss_avg, train_taski_loss_avg,
                         taski, step=1, val_choose="TF")
                train_loss_avg.reset()
                train_taski_loss_avg.reset()
            if nxt is None and iteration < n_iter:
                nxt = fetch()

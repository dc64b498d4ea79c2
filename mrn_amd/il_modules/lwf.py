"""LwF learner (reference il_modules/lwf.py:26-114): tasks > 0 add a knowledge-distillation term against the frozen
previous network: loss = 3 * KD(T=2 on the old classes) + loss_clf."""
import contextlib
import time

import torch

from .. import functional as Fn
from .. import ops
from ..tools.utils import Averager
from .base import BaseLearner

T = 2
lamda = 3


class LwF(BaseLearner):
    kd_weight = lamda

    def after_task(self):
        self.model = self.model.module
        self._old_network = self.model.copy().freeze()
        self._known_classes = self._total_classes

    def _old_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = ops.aux_stream(0, self.device)
        return self._side

    def kd_step(self, image, labels):
        """one iteration of lwf.py:52-95"""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        ctc = "CTC" in self.opt.Prediction
        text = None if ctc else labels_index[:, :-1]
        # the frozen + eval-mode copy (Model.freeze(), model.py:191-197) does not depend on the network being trained: its forward
        # runs on a side stream next to the new network's (same results, only the launch order changes)
        side = self._old_stream() if image.is_cuda else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            old = self._old_network(image, text, True)["predict"]
        preds = self.model(image, text, True)["predict"]
        if side is not None:
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            old.record_stream(main)
            image.record_stream(side)
            if text is not None:
                text.record_stream(side)
        loss_clf = self.criterion(preds, labels_index, labels_length)
        loss_kd = Fn.kd_loss(preds, old, 0 if ctc else 1, self._known_classes, T)          # [:, start_index:known] (:81-85)
        loss = self.kd_weight * loss_kd + loss_clf
        self.backward_and_step(loss)
        return loss, loss_kd

    def _update_representation(self, start_iter, taski, train_loader, valid_loader):
        train_loader.get_dataset(taski, memory=self.opt.memory)             # lwf.py:37 (on top of what _train already loaded)
        avg = Averager()
        start_time = time.time()
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image, labels = train_loader.get_batch()
            loss, _ = self.kd_step(image.to(self.device), labels)
            avg.add(loss.detach())
            self.end_iteration(iteration)
            if iteration % self.opt.val_interval == 0 or iteration == 1:
                self.val(valid_loader, self.opt, -1, start_time, iteration, avg, None, taski)
                avg.reset()

"""DER learner (reference il_modules/der.py:28-348): a new extractor per task, old extractors frozen and in eval mode,
loss = loss_clf (the auxiliary loss is computed and logged but NOT added, reference :264-265), weight_align after
every task > 0."""
import time

from .. import parallel
from ..modules.model import DERNet
from ..tools.utils import Averager
from .base import BaseLearner


class DER(BaseLearner):
    def __init__(self, opt):
        super().__init__(opt)
        self.model = DERNet(opt)

    def after_task(self):
        self.model = self.model.module
        self._known_classes = self._total_classes

    def model_eval_and_train(self, taski):
        self.model.train()
        self.model.module.model[-1].train()
        for i in range(taski):
            self.model.module.model[i].eval()

    def _expand(self):
        net = self.model.module if isinstance(self.model, parallel.ReplicaDataParallel) else self.model
        net.update_fc(self.opt.hidden_size, self._total_classes)
        net.build_prediction(self.opt, self._total_classes)
        net.build_aux_prediction(self.opt, self._total_classes)
        return net

    def change_model(self):
        self._wrap(self._expand())

    def build_model(self):
        self.model = self._expand()
        self._reference_init()
        self._wrap(self.model)

    def incremental_train(self, taski, character, train_loader, valid_loader):
        self.character = character
        self.converter = self.build_converter()
        valid = valid_loader.create_dataset()
        if taski > 0:
            self.change_model()
        else:
            self.criterion = self.build_criterion()
            self.build_model()
        for i in range(taski):                               # der.py:101-104
            for p in self.model.module.model[i].parameters():
                p.requires_grad = False
        self.build_optimizer(self.count_param())
        if self.opt.start_task > taski:                      # resume (der.py:112-129)
            if taski > 0:
                self.load_task_data(train_loader, taski)
            self.load_checkpoint(self.checkpoint_path(taski))
        else:
            print("Task {} start training for model ------{}------".format(taski, self.opt.exp_name))
            self._train(0, taski, train_loader, valid)

    def _train(self, start_iter, taski, train_loader, valid_loader):
        if taski == 0:
            self._update(start_iter, taski, train_loader, valid_loader)
        else:
            self.load_task_data(train_loader, taski)
            self.model_eval_and_train(taski)
            self._update(start_iter, taski, train_loader, valid_loader)
            self.model.module.weight_align(self._total_classes - self._known_classes)       # der.py:148

    def prefetch_frozen(self, image):
        """issue the frozen extractors' forward of a FUTURE batch (DERNet.frozen_prefetch); pass the result to
        der_step(..., prefetched=...).  Only the launch order changes, not the results."""
        return self.model.module.frozen_prefetch(image)

    def der_step(self, image, labels, prefetched=None):
        """one iteration of der.py:226-271"""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        if "CTC" in self.opt.Prediction:
            output = self.model(image, frozen=prefetched)
        else:
            output = self.model(image, labels_index[:, :-1], frozen=prefetched)
        loss_clf = self.criterion(output["logits"], labels_index, labels_length)
        loss_aux = self.criterion(output["aux_logits"].detach(), labels_index, labels_length)   # logged only (:264-265)
        self.backward_and_step(loss_clf)
        return loss_clf, loss_aux

    def _update(self, start_iter, taski, train_loader, valid_loader):
        avg, clf_avg, aux_avg = Averager(), Averager(), Averager()
        start_time = time.time()
        # one batch of look-ahead: the frozen extractors of batch n+1 run on a side stream while batch n trains the newest one
        nxt = None
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            if nxt is None:
                image, labels = train_loader.get_batch()
                image, pre = image.to(self.device), None
            else:
                image, labels, pre = nxt
            nxt = None
            if taski > 0 and iteration < self.opt.num_iter and iteration % self.opt.val_interval != 0 and iteration != 1:
                ni, nl = train_loader.get_batch()
                ni = ni.to(self.device)
                nxt = (ni, nl, self.prefetch_frozen(ni))
            loss, aux = self.der_step(image, labels, prefetched=pre)
            avg.add(loss.detach())
            clf_avg.add(loss.detach())
            aux_avg.add(aux.detach())
            self.end_iteration(iteration)
            if iteration % self.opt.val_interval == 0 or iteration == 1:
                self.val(valid_loader, self.opt, -1, start_time, iteration, avg, None, taski)
                print(f"CLF_loss: {float(clf_avg.val()):0.5f} , Aux_loss: {float(aux_avg.val()):0.5f}")
                for a in (avg, clf_avg, aux_avg):
                    a.reset()

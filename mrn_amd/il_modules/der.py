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
        self.model = parallel.ReplicaDataParallel(self._expand()).to(self.device)
        self.model.train()

    def build_model(self):
        self.model = self._expand()
        self._reference_init()
        self.model = parallel.ReplicaDataParallel(self.model).to(self.device)
        self.model.train()

    def incremental_train(self, taski, character, train_loader, valid_loader):
        self.character = character
        self.converter = self.build_converter()
        valid = valid_loader.create_dataset()
        if taski > 0:
            self.change_model()
        else:
            self.criterion = self.build_criterion()
            self.build_model()
        for i in range(taski):
            for p in self.model.module.model[i].parameters():
                p.requires_grad = False
        self.build_optimizer(self.count_param())
        self._train(0, taski, train_loader, valid)

    def _train(self, start_iter, taski, train_loader, valid_loader):
        if taski == 0:
            self._update(start_iter, taski, train_loader, valid_loader)
        else:
            train_loader.get_dataset(taski, memory=self.opt.memory)
            self.model_eval_and_train(taski)
            self._update(start_iter, taski, train_loader, valid_loader)
            self.model.module.weight_align(self._total_classes - self._known_classes)

    def der_step(self, image, labels):
        """one iteration of der.py:226-271"""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        if "CTC" in self.opt.Prediction:
            output = self.model(image)
        else:
            output = self.model(image, labels_index[:, :-1])
        loss_clf = self.criterion(output["logits"], labels_index, labels_length)
        loss_aux = self.criterion(output["aux_logits"].detach(), labels_index, labels_length)   # logged only
        self.optimizer_step(loss_clf)
        return loss_clf, loss_aux

    def _update(self, start_iter, taski, train_loader, valid_loader):
        avg, aux_avg = Averager(), Averager()
        start_time, best = time.time(), -1
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image, labels = train_loader.get_batch()
            loss, aux = self.der_step(image.to(self.device), labels)
            avg.add(loss.detach())
            aux_avg.add(aux.detach())
            if iteration % self.opt.val_interval == 0 or iteration == 1:
                best = self.val(valid_loader, self.opt, best, start_time, iteration, avg, None, taski, None, "val")
                avg.reset()
                aux_avg.reset()

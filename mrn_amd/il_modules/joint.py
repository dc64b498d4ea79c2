"""Joint-training learner (reference il_modules/joint.py:9-105): the canonical loop of BaseLearner on the union of all
tasks' data, with the per-dataset test pass run at every validation interval."""
import time

from ..tools.utils import Averager
from .base import BaseLearner


class JointLearner(BaseLearner):
    def incremental_train(self, taski, character, train_loader, valid_loader, AlignCollate_valid=None, valid_datas=None):
        self.character = character
        self.converter = self.build_converter()
        try:
            valid = valid_loader.create_list_dataset(valid_datas=valid_datas)      # joint.py:16
        except TypeError:
            valid = valid_loader.create_list_dataset()
        if taski > 0:
            self.change_model()
        else:
            self.criterion = self.build_criterion()
            self.build_model()
        self.build_optimizer(self.count_param())
        return self._init_train(0, taski, train_loader, valid, AlignCollate_valid, valid_datas)

    def _init_train(self, start_iter, taski, train_loader, valid_loader, AlignCollate_valid=None, valid_datas=None):
        train_loss_avg = Averager()
        best_scores, ned_scores = [], []
        start_time = time.time()
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image_tensors, labels = train_loader.get_batch()
            loss = self.train_step(image_tensors.to(self.device), labels)
            train_loss_avg.add(loss.detach())
            self.end_iteration(iteration)
            if iteration % self.opt.val_interval == 0 or iteration == 1:
                self.val(valid_loader, self.opt, -1, start_time, iteration, train_loss_avg, None, taski)
                if iteration != 1 and valid_datas is not None:
                    # the reference indexes its score lists by task; joint training reports one running entry
                    scores, neds = self.test(AlignCollate_valid, valid_datas, [0.0] * taski, [0.0] * taski, taski)
                    best_scores, ned_scores = scores[taski:], neds[taski:]
                    self.model.train()
                train_loss_avg.reset()
        return best_scores, ned_scores

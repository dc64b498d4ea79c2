"""WA learner (reference il_modules/wa.py:29-116): LwF-style distillation with weight 2 against the frozen previous
network, plus weight alignment of the newest classifier rows (mean row norm of the new classes scaled to the old
classes', modules/model.py:166-174) at the end of every incremental task and again in after_task()."""
from .lwf import LwF

T = 2


class WA(LwF):
    kd_weight = 2          # wa.py:88: loss = loss_clf + 2 * loss_kd

    def __init__(self, opt):
        super().__init__(opt)
        self.taski = 0

    def after_task(self):
        if self.taski > 0:                                  # wa.py:34-36
            self.model.module.weight_align(self._total_classes - self._known_classes)
        super().after_task()

    def _update_representation(self, start_iter, taski, train_loader, valid_loader):
        self.taski = taski
        super()._update_representation(start_iter, taski, train_loader, valid_loader)
        self.model.module.weight_align(self._total_classes - self._known_classes)          # wa.py:110

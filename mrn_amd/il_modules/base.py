"""BaseLearner: the plugin base of the incremental-learning strategies (reference il_modules/base.py:26-467), with
the reference's call contract (`incremental_train(taski, character, train_loader, valid_loader)`, `test(...)`,
`after_task()`) and attribute names, driving the HIP-backed model containers.

MI355X-first differences: one process per GPU with a flat-buffer Adam and one RCCL all-reduce per step instead of
torch.nn.DataParallel; the loss is a fused log-softmax+CTC / cross-entropy kernel instead of three torch ops.
"""
import os
import time

import torch
import torch.nn.init as init

from .. import functional as Fn
from .. import parallel
from ..modules.model import Model
from ..optim import FlatAdam, OneCycle
from ..test import validation
from ..tools.utils import AttnLabelConverter, Averager, CTCLabelConverter, adjust_learning_rate


class Criterion:
    """Callable loss with the learners' calling convention: criterion(preds [B,T,C], labels_index, labels_length)."""

    def __init__(self, prediction, pad_index=None):
        self.prediction, self.pad_index = prediction, pad_index

    def __call__(self, preds, labels_index, labels_length=None):
        if "CTC" in self.prediction:                       # log_softmax + CTCLoss(mean, zero_infinity), base.py:131
            if preds.stride(-1) != 1 or preds.stride(0) != preds.shape[1] * preds.stride(1):
                preds = preds.contiguous()
            return Fn.ctc_loss(preds, labels_index, labels_length)
        target = labels_index[:, 1:]                       # without [SOS]; CrossEntropyLoss(ignore_index=[PAD]), :134
        return Fn.cross_entropy(preds, target, self.pad_index)

    def to(self, device):
        return self


class BaseLearner(object):
    def __init__(self, opt):
        self._cur_task = -1
        self._known_classes = 0
        self._total_classes = 0
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.opt = opt
        self.character = None
        self.optimizer = None
        self.scheduler = None
        self.criterion = None
        self.converter = None
        self.memory_index = []
        self._old_network = None
        self.model = Model(opt)
        self.iteration = 0

    # -- construction ------------------------------------------------------------------------------------
    def _reference_init(self):
        """weight initialisation of task 0 (base.py:53-65): kaiming for weights, 0 for biases, 1 for norm scales"""
        for name, param in self.model.named_parameters():
            if "localization_fc2" in name:
                print(f"Skip {name} as it is already initialized")
                continue
            try:
                if "bias" in name:
                    init.constant_(param, 0.0)
                elif "weight" in name:
                    init.kaiming_normal_(param)
            except Exception:                      # 1-D weights (BatchNorm / LayerNorm)
                if "weight" in name:
                    param.data.fill_(1)

    def build_model(self):
        self.model.update_fc(self.opt.hidden_size, self._total_classes)
        self.model.build_prediction(self.opt, self._total_classes)
        self._reference_init()
        self.model = parallel.ReplicaDataParallel(self.model).to(self.device)
        self.model.train()

    def count_param(self):
        params, total = [], 0
        for p in filter(lambda p: p.requires_grad, self.model.parameters()):
            params.append(p)
            total += p.numel()
        print("Trainable params num : ", total)
        return params

    def build_optimizer(self, filtered_parameters, scale=1.0, total_steps=None):
        if self.opt.optimizer != "adam":
            raise NotImplementedError(f"optimizer '{self.opt.optimizer}': only Adam (the shipped configs) runs on the HIP path")
        self.optimizer = FlatAdam(filtered_parameters, lr=self.opt.lr * scale)
        parallel.broadcast_parameters(self.optimizer.flat)
        self.opt_step = 0
        if "super" in self.opt.schedule:
            self.scheduler = OneCycle(self.opt.lr * scale, total_steps or self.opt.num_iter)
        else:
            self.scheduler = None
        self.write_log(f"FlatAdam(lr={self.opt.lr * scale}, n={self.optimizer.flat.numel()})\n")

    def build_converter(self):
        if "CTC" in self.opt.Prediction:
            converter = CTCLabelConverter(self.character)
        else:
            converter = AttnLabelConverter(self.character)
            self.sos_token_index = converter.dict["[SOS]"]
            self.eos_token_index = converter.dict["[EOS]"]
        self._total_classes = len(converter.character)
        return converter

    def build_criterion(self, reduction="mean"):
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' is used on the path")
        pad = None if "CTC" in self.opt.Prediction else self.converter.dict["[PAD]"]
        return Criterion(self.opt.Prediction, pad)

    def change_model(self):
        if isinstance(self.model, parallel.ReplicaDataParallel):
            self.model = self.model.module
        self.model.update_fc(self.opt.hidden_size, self._total_classes)
        self.model.build_prediction(self.opt, self._total_classes)
        self.model = parallel.ReplicaDataParallel(self.model).to(self.device)
        self.model.train()

    # -- the optimiser step shared by every learner (base.py:255-269) -------------------------------------
    def optimizer_step(self, loss):
        self.optimizer.zero_grad()
        loss.backward()
        parallel.average_gradients(self.optimizer.grad)
        if self.scheduler is not None:
            lr = self.scheduler.lr_at(self.opt_step)
        else:
            lr = self.optimizer.lr
        self.optimizer.step(lr=lr, max_norm=self.opt.grad_clip)
        self.opt_step += 1

    def incremental_train(self, taski, character, train_loader, valid_loader):
        self.character = character
        self.converter = self.build_converter()
        if taski > 0:
            self.change_model()
        else:
            self.criterion = self.build_criterion()
            self.build_model()
        self.build_optimizer(self.count_param())
        self._train(0, taski, train_loader, valid_loader)

    def _train(self, start_iter, taski, train_loader, valid_loader):
        print("Task {} start training for model ------{}------".format(taski, self.opt.exp_name))
        self._init_train(start_iter, taski, train_loader, valid_loader.create_dataset())

    def train_step(self, image, labels):
        """one iteration of the canonical loop (base.py:226-264): forward, loss, backward, clip, Adam, schedule"""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        if "CTC" in self.opt.Prediction:
            preds = self._forward_train(image, None)
        else:
            preds = self._forward_train(image, labels_index[:, :-1])
        loss = self.criterion(preds, labels_index, labels_length)
        self.optimizer_step(loss)
        return loss

    def _forward_train(self, image, text):
        out = self.model(image, text)
        return out["predict"] if "predict" in out else out["logits"]

    def _init_train(self, start_iter, taski, train_loader, valid_loader, cross=False):
        train_loss_avg = Averager()
        start_time = time.time()
        best_score = -1
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image_tensors, labels = train_loader.get_batch()
            loss = self.train_step(image_tensors.to(self.device), labels)
            train_loss_avg.add(loss.detach())
            if self.scheduler is None:
                adjust_learning_rate(self.optimizer, iteration, self.opt)
            if iteration % self.opt.val_interval == 0 or iteration == self.opt.num_iter:
                self.val(valid_loader, self.opt, best_score, start_time, iteration, train_loss_avg, None, taski, 0, "FF")
                train_loss_avg.reset()

    # -- evaluation / bookkeeping -------------------------------------------------------------------------
    def val(self, valid_loader, opt, best_score, start_time, iteration, train_loss_avg, train_taski_loss_avg, taski,
            step=0, val_choose="val"):
        self.model.eval()
        with torch.no_grad():
            (valid_loss, current_score, ned_score, preds, confidence_score, labels, infer_time,
             length_of_data) = validation(self.model, self.criterion, valid_loader, self.converter, opt, val_choose=val_choose)
        self.model.train()
        if current_score > best_score:
            best_score = current_score
            self.save_checkpoint(taski, step)
        lr = self.optimizer.param_groups[0]["lr"]
        log = (f"\n[{iteration}/{opt.num_iter}] Train_loss_clf: {float(train_loss_avg.val()):0.5f}, Valid_loss: {valid_loss:0.5f}\n"
               f'{"":9s}Current_score: {current_score:0.2f}, Ned_score: {ned_score:0.2f}\n'
               f'{"":9s}Current_lr: {lr:0.7f}, Best_score: {best_score:0.2f}\n')
        if train_taski_loss_avg is not None:
            log += f'{"":9s}Train_taski_loss: {float(train_taski_loss_avg.val()):0.5f}\n'
        print(log)
        self.write_log(log + "\n")
        return best_score

    def checkpoint_path(self, taski, step=None):
        name = self.opt.lan_list[taski]
        tail = f"{name}_{taski}_best_score.pth" if step is None else f"{name}_{taski}_{step}_best_score.pth"
        return f"./saved_models/{self.opt.exp_name}/{tail}"

    def save_checkpoint(self, taski, step=None):
        """reference-format checkpoint: state_dict of the wrapped model (`module.` keys), base.py:323-332"""
        if parallel.world_size() > 1 and torch.distributed.get_rank() != 0:
            return
        path = self.checkpoint_path(taski, None if step == 0 and type(self) is BaseLearner else step)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.model.state_dict(), path)

    def test(self, AlignCollate_valid, valid_datas, best_scores, ned_scores, taski, val_choose="test"):
        """valid_datas: iterable of evaluation loaders (the LMDB hierarchy of the reference is out of scope)"""
        accs, neds = [], []
        for loader in valid_datas:
            self.model.eval()
            with torch.no_grad():
                _, acc, ned, *_ = validation(self.model, self.criterion, loader, self.converter, self.opt, val_choose=val_choose)
            accs.append(round(acc, 2))
            neds.append(round(ned, 2))
        self.model.train()
        best_scores.append(round(sum(accs) / max(len(accs), 1), 2))
        ned_scores.append(round(sum(neds) / max(len(neds), 1), 2))
        self.write_log(f"Task {taski} Test Average Incremental Accuracy: {best_scores[taski]}\n")
        return best_scores, ned_scores

    def after_task(self):
        self.model = self.model.module
        self._known_classes = self._total_classes
        self._old_network = self.model.copy().freeze()

    def write_log(self, line):
        d = f"./saved_models/{self.opt.exp_name}"
        try:
            os.makedirs(d, exist_ok=True)
            with open(f"{d}/log_train.txt", "a") as f:
                f.write(line)
        except OSError:
            pass

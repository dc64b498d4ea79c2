"""BaseLearner: the plugin base of the incremental-learning strategies (reference il_modules/base.py:26-467), with
the reference's call contract (`incremental_train(taski, character, train_loader, valid_loader)`, `test(...)`,
`after_task()`) and attribute names, driving the HIP-backed model containers.

MI355X-first differences: one process per GPU with a flat-buffer optimiser and bucketed RCCL all-reduces overlapped
with backward instead of torch.nn.DataParallel; the loss is a fused log-softmax+CTC / cross-entropy kernel instead of
three torch ops.
"""
import os
import time

import numpy as np
import torch
import torch.nn.init as init

from .. import functional as Fn
from .. import ops, parallel
from ..modules.model import Model
from ..optim import FlatAdadelta, FlatAdam, FlatSGD, OneCycle
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
        loss = Fn.cross_entropy(preds, target, self.pad_index)
        if torch.is_grad_enabled():                        # N > 1: the mean over the GLOBAL batch's valid targets, as DataParallel
            w = parallel.global_mean_weight((target != self.pad_index).sum())       # computes it on the gathered outputs
            if w is not None:
                loss = loss * w.view(loss.shape)
        return loss

    def to(self, device):
        return self


class BaseLearner(object):
    checkpoint_has_step = False        # MRN names its checkpoints {lan}_{taski}_{step}_best_score.pth (mrn.py:414)

    def __init__(self, opt):
        self._cur_task = -1
        self._known_classes = 0
        self._total_classes = 0
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.opt = opt
        self.character = None
        self.optimizer = None
        self.scheduler = None
        self.reducer = None
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

    def _wrap(self, net):
        """replaces torch.nn.DataParallel(self.model).to(device) (base.py:68): one replica per process; every replica starts
        from rank 0's parameters AND buffers (DataParallel re-broadcasts them from GPU 0 every iteration)"""
        self.model = parallel.ReplicaDataParallel(net).to(self.device)
        parallel.broadcast_module(self.model)
        self.model.train()

    def build_model(self):
        self.model.update_fc(self.opt.hidden_size, self._total_classes)
        self.model.build_prediction(self.opt, self._total_classes)
        self._reference_init()
        self._wrap(self.model)

    def count_param(self):
        params, total = [], 0
        for p in filter(lambda p: p.requires_grad, self.model.parameters()):
            params.append(p)
            total += p.numel()
        print("Trainable params num : ", total)
        return params

    def build_optimizer(self, filtered_parameters, scale=1.0, total_steps=None, optimizer=None, schedule=None):
        """base.py:72-114 (and MRN.build_custom_optimizer, mrn.py:52-94): Adam / SGD / Adadelta over the flat buffers;
        "super" schedules = OneCycle (momentum cycled for SGD), anything else = the stepwise adjust_learning_rate() of
        tools/utils.py:169-178 (the reference also constructs a MultiStepLR there but never steps it)."""
        name = optimizer or self.opt.optimizer
        schedule = self.opt.schedule if schedule is None else schedule
        lr = self.opt.lr * scale
        if name == "sgd":
            self.optimizer = FlatSGD(filtered_parameters, lr, momentum=self.opt.sgd_momentum, weight_decay=self.opt.sgd_weight_decay)
        elif name == "adadelta":
            self.optimizer = FlatAdadelta(filtered_parameters, lr, rho=self.opt.rho, eps=self.opt.eps)
        elif name == "adam":
            self.optimizer = FlatAdam(filtered_parameters, lr=lr)
        else:
            raise ValueError(f"unknown optimizer '{name}' (sgd | adadelta | adam)")
        parallel.broadcast_parameters(self.optimizer.flat, params=self.optimizer.params)
        if self.reducer is not None:
            self.reducer.close()
        self.reducer = parallel.BucketedAllReduce(self.optimizer) if parallel.world_size() > 1 else None
        self.opt_step = 0
        if "super" in schedule:
            self.scheduler = OneCycle(lr, total_steps or self.opt.num_iter, cycle_momentum=(name == "sgd"))
        else:
            self.scheduler = None
        self.write_log(f"{type(self.optimizer).__name__}(lr={lr}, n={self.optimizer.flat.numel()})\n")

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
        self._wrap(self.model)

    # -- the optimiser step shared by every learner (base.py:255-269) -------------------------------------
    def backward_and_step(self, loss, after_reduce=None):
        """zero_grad, backward (gradient buckets all-reduced while it runs), clip, update, schedule.  `after_reduce`: hook that
        adds replica-independent terms to the averaged flat gradient (the EWC penalty depends on the parameters only)."""
        self.optimizer.zero_grad()
        # parameter gradients of the trained layers: second stream, straight into the flat gradient (ops.direct_gradients); with N > 1
        # ranks the buckets count autograd's post-accumulate hooks AND the side stream's completions (notify), so the data-parallel
        # step runs the same schedule as the single-GPU one
        if self.reducer is not None:
            self.reducer.begin()
            with ops.direct_gradients(notify=self.reducer.param_ready, roots=(loss,)):
                loss.backward()
            self.reducer.finish()
        else:
            with ops.direct_gradients():
                loss.backward()
        if after_reduce is not None:
            after_reduce()
        momentum = None
        if self.scheduler is not None:                     # scheduler.step() after every optimizer.step()
            lr = self.scheduler.lr_at(self.opt_step)
            momentum = self.scheduler.momentum_at(self.opt_step)
        else:                                              # adjust_learning_rate() wrote param_groups[0]["lr"]
            lr = self.optimizer.param_groups[0]["lr"]
        self.optimizer.step(lr=lr, max_norm=self.opt.grad_clip, momentum=momentum)
        self.opt_step += 1
        ops.prepack_trained()             # next step's weight operands of the trained convolutions, on the side stream

    optimizer_step = backward_and_step

    def end_iteration(self, iteration):
        """the non-"super" schedule of every reference loop (base.py:266-269): stepwise decay by iteration count"""
        if self.scheduler is None:
            adjust_learning_rate(self.optimizer, iteration, self.opt)

    def incremental_train(self, taski, character, train_loader, valid_loader):
        self.character = character
        self.converter = self.build_converter()
        valid_loader = valid_loader.create_dataset()
        if taski > 0:
            self.change_model()
        else:
            self.criterion = self.build_criterion()
            self.build_model()
        self.build_optimizer(self.count_param())
        if self.opt.start_task > taski:                    # resume: skip training, load the task's checkpoint (base.py:178-195)
            if taski > 0:
                self.load_task_data(train_loader, taski)
            self.load_checkpoint(self.checkpoint_path(taski))
        else:
            print("Task {} start training for model ------{}------".format(taski, self.opt.exp_name))
            self._train(0, taski, train_loader, valid_loader)

    def load_task_data(self, train_loader, taski):
        """what every task > 0 does before training (base.py:209-213): rehearsal memory or the plain task dataset"""
        if self.opt.memory is not None:
            self.build_rehearsal_memory(train_loader, taski)
        else:
            train_loader.get_dataset(taski, memory=self.opt.memory)

    def _train(self, start_iter, taski, train_loader, valid_loader):
        if taski == 0:
            self._init_train(start_iter, taski, train_loader, valid_loader)
        else:
            self.load_task_data(train_loader, taski)
            self._update_representation(start_iter, taski, train_loader, valid_loader)

    def train_step(self, image, labels):
        """one iteration of the canonical loop (base.py:226-264): forward, loss, backward, clip, update, schedule"""
        labels_index, labels_length = self.converter.encode(labels, batch_max_length=self.opt.batch_max_length)
        if "CTC" in self.opt.Prediction:
            preds = self._forward_train(image, None)
        else:
            preds = self._forward_train(image, labels_index[:, :-1])
        loss = self.criterion(preds, labels_index, labels_length)
        self.backward_and_step(loss)
        return loss

    def _forward_train(self, image, text):
        out = self.model(image, text, True)
        return out["predict"] if "predict" in out else out["logits"]

    def _init_train(self, start_iter, taski, train_loader, valid_loader):
        train_loss_avg = Averager()
        start_time = time.time()
        best_score = -1
        for iteration in range(start_iter + 1, self.opt.num_iter + 1):
            image_tensors, labels = train_loader.get_batch()
            loss = self.train_step(image_tensors.to(self.device), labels)
            train_loss_avg.add(loss.detach())
            self.end_iteration(iteration)
            if iteration % self.opt.val_interval == 0 or iteration == 1:       # base.py:273
                self.val(valid_loader, self.opt, best_score, start_time, iteration, train_loss_avg, None, taski)
                train_loss_avg.reset()

    def _update_representation(self, start_iter, taski, train_loader, valid_loader):
        self._init_train(start_iter, taski, train_loader, valid_loader)

    # -- rehearsal memory (base.py:278-302): learner-side index bookkeeping, the data manager does the sampling ------------
    def memory_per_task(self, taski):
        return int(self.opt.memory_num / taski)

    def build_rehearsal_memory(self, train_loader, taski):
        num_i = self.memory_per_task(taski)
        self.build_random_current_memory(num_i, taski, train_loader)
        if len(self.memory_index) != 0 and len(self.memory_index) * len(self.memory_index[0]) > self.opt.memory_num:
            self.reduce_samplers(taski, taski_num=num_i)
        train_loader.get_dataset(taski, memory=self.opt.memory, index_list=self.memory_index)
        print("Is using rehearsal memory, has {} prev datasets, each has {}\n".format(len(self.memory_index), self.memory_index[0].size))

    def build_random_current_memory(self, taski_num, taski, train_loader):
        """taski_num random sample indices of the PREVIOUS task's dataset join the memory"""
        prev_loader, len_data = train_loader.rehearsal_prev_model(taski)
        index_list = np.random.choice(range(len_data), taski_num, replace=False)
        self.memory_index.append(index_list)

    def reduce_samplers(self, taski, taski_num):
        for i in range(taski):
            self.memory_index[i] = self.memory_index[i][:taski_num]
            print("----using memory {}".format(self.memory_index[i].size))

    # -- evaluation / bookkeeping -------------------------------------------------------------------------
    def val(self, valid_loader, opt, best_score, start_time, iteration, train_loss_avg, train_taski_loss_avg, taski,
            step=None, val_choose="val"):
        self.model.eval()
        with torch.no_grad():
            (valid_loss, current_score, ned_score, preds, confidence_score, labels, infer_time,
             length_of_data) = validation(self.model, self.criterion, valid_loader, self.converter, opt, val_choose=val_choose)
        self.model.train()
        # (the reference loops never read val()'s result back, so their best_score stays -1 and every validation whose
        #  score beats -1 overwrites the checkpoint: base.py:323-332 with :272-274)
        if current_score > best_score:
            best_score = current_score
            self.save_checkpoint(taski, step)
        ned_score = 0.0 if ned_score is None else ned_score
        lr = self.optimizer.param_groups[0]["lr"]
        log = (f"\n[{iteration}/{opt.num_iter}] Train_loss: {float(train_loss_avg.val()):0.5f}, Valid_loss: {float(valid_loss):0.5f}\n"
               f'{"":9s}Current_score: {current_score:0.2f}, Ned_score: {ned_score:0.2f}\n'
               f'{"":9s}Current_lr: {lr:0.7f}, Best_score: {best_score:0.2f}\n'
               f'{"":9s}Infer_time: {infer_time:0.2f},     Elapsed_time: {time.time() - start_time:0.2f}\n')
        if train_taski_loss_avg is not None:
            log += f'{"":9s}Train_taski_loss: {float(train_taski_loss_avg.val()):0.5f}\n'
        skipped = self.optimizer.skipped_steps() if hasattr(self.optimizer, "skipped_steps") else 0
        if skipped:     # (the reference would have gone NaN and been noticed; here an overflowed gradient skips its step, csrc/optim.hip)
            log += f'{"":9s}Optimiser steps SKIPPED for a non-finite gradient norm so far: {skipped}\n'
        dashed = "-" * 80
        log += f'{dashed}\n{"Ground Truth":25s} | {"Prediction":25s} | Confidence Score & T/F\n{dashed}\n'
        for gt, pred, confidence in zip(labels[:5], preds[:5], confidence_score[:5]):
            if "Attn" in opt.Prediction:                   # (as the reference log does, base.py:351-353)
                gt = gt[: gt.find("[EOS]")]
                pred = pred[: pred.find("[EOS]")]
            log += f"{gt:25s} | {pred:25s} | {float(confidence):0.4f}\t{str(pred == gt)}\n"
        log += dashed
        print(log)
        self.write_log(log + "\n")
        return best_score

    def checkpoint_path(self, taski, step=None):
        name = self.opt.lan_list[taski]
        tail = f"{name}_{taski}_{step}_best_score.pth" if (self.checkpoint_has_step and step is not None) else f"{name}_{taski}_best_score.pth"
        return f"./saved_models/{self.opt.exp_name}/{tail}"

    def save_checkpoint(self, taski, step=None):
        """reference-format checkpoint: state_dict of the wrapped model (`module.` keys), base.py:323-332; rank 0 writes"""
        if parallel.rank() != 0:
            return
        path = self.checkpoint_path(taski, step)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        torch.save(self.model.state_dict(), tmp)
        os.replace(tmp, path)                    # readers (other ranks, a resumed run) never see a partly written file

    def load_checkpoint(self, path):
        """reference `self.model.load_state_dict(torch.load(path), strict=True)` (base.py:193, :375); both TPS-buffer key
        flavours of reference checkpoints load (modules/transformation.py GridGenerator._load_from_state_dict)"""
        sd = torch.load(path, map_location=self.device)
        self.model.load_state_dict(sd, strict=True)
        print("Task load checkpoint from {}.".format(path))

    def test(self, AlignCollate_valid, valid_datas, best_scores, ned_scores, taski, val_choose="test", step=None):
        """base.py:363-423: reload the task's saved checkpoint, evaluate every test set.  valid_datas: iterable of evaluation
        loaders, or LMDB roots when an LMDB reader is available (mrn_amd.data.dataset.hierarchical_dataset)"""
        path = self.checkpoint_path(taski, step)
        parallel.barrier()                           # rank 0's last save_checkpoint() of the task is on disk before any rank looks
        if os.path.exists(path):
            self.load_checkpoint(path)
        accs, neds = [], []
        for loader in valid_datas:
            if isinstance(loader, str):
                from ..data.data_manage import evaluation_loader
                loader = evaluation_loader(loader, self.opt, AlignCollate_valid)
            self.model.eval()
            with torch.no_grad():
                _, acc, ned, *_ = validation(self.model, self.criterion, loader, self.converter, self.opt, val_choose=val_choose)
            accs.append(round(acc, 2))
            neds.append(round(ned, 2))
        self.model.train()
        if (taski + 1) * 2 == len(accs):                   # MLT17 / MLT19 pairs per task (base.py:399-405, double_write :425-436)
            s17 = round(sum(accs[0::2][: taski + 1]) / (taski + 1), 2)
            s19 = round(sum(accs[1::2][: taski + 1]) / (taski + 1), 2)
            best_scores.append(s17)
            ned_scores.append(s19)
            log = f"Task {taski} Avg Incremental Acc:  17: {s17}    19: {s19}\n"
        else:
            best_scores.append(round(sum(accs) / max(len(accs), 1), 2))
            ned_scores.append(round(sum(neds) / max(len(neds), 1), 2))
            log = (f"Task {taski} Test Average Incremental Accuracy: {best_scores[taski]} \n Task {taski} Incremental Accuracy: {accs}\n"
                   f" ned_acc: {neds}\n")
        print(log)
        self.write_log(log)
        return best_scores, ned_scores

    def after_task(self):
        self.model = self.model.module
        self._old_network = self.model.copy().freeze()
        self._known_classes = self._total_classes

    def write_log(self, line):
        if parallel.rank() != 0:
            return
        d = f"./saved_models/{self.opt.exp_name}"
        try:
            os.makedirs(d, exist_ok=True)
            with open(f"{d}/log_train.txt", "a") as f:
                f.write(line)
        except OSError:
            pass

#!/usr/bin/env python3
"""Training driver with the reference's contract (reference tiny_train.py:195-294, 407-460):

    python -m mrn_amd.tiny_train --config <config.py> [--synthetic]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m mrn_amd.tiny_train --config ...

The config is the reference's mmcv-style Python file (sections `common`, `model`, `train`, `optimizer`, merged into one
namespace exactly as tiny_train.py:413-422 does; as there, no other command-line flag overrides it); the reference's own
config/*.py files load unchanged.  `train(opt, log)` picks the learner by opt.il, builds the Dataset_Manager / Val_Dataset,
and runs every task through `learner.incremental_train(taski, opt.character, train_loader, valid_loader)`,
`learner.test(...)`, `learner.after_task()`.

MI355X-first differences: one process per GPU (torchrun) instead of DataParallel inside one process -- every rank runs this
same loop on its own shard of each batch, gradients meet in the learners' bucketed RCCL all-reduce; `--synthetic` swaps the
LMDB datasets for device-generated crops (the benchmark's data) so that the whole driver runs without any dataset on disk.
"""
import argparse
import os
import random
import runpy
import sys
import types

import numpy as np
import torch

from . import parallel
from .il_modules.base import BaseLearner
from .il_modules.der import DER
from .il_modules.ewc import EWC
from .il_modules.joint import JointLearner
from .il_modules.lwf import LwF
from .il_modules.mrn import MRN
from .il_modules.wa import WA
from .tools.utils import host_cpu_budget

LEARNERS = {"lwf": LwF, "wa": WA, "ewc": EWC, "der": DER, "mrn": MRN, "joint_mix": JointLearner, "joint_loader": JointLearner}


def write_data_log(line):
    if parallel.rank() == 0:
        with open("data_any.txt", "a+") as log:
            log.write(line)


def load_dict(path, char):
    """<path>/dict.txt: one character per line; `char` accumulates the characters of all tasks so far, in first-seen order
    (reference tiny_train.py:37-53) -> (cumulative character list, char)"""
    with open(path + "/dict.txt") as f:
        ch_list = [line.strip("\n") for line in f]
    for ch in ch_list:
        if char.get(ch, None) is None:
            char[ch] = 1
    character = list(char.keys())
    print("dict has {} number characters\n".format(len(character)))
    return character, char


def load_config(path):
    """merge the sections of an mmcv-style config file into one namespace, in the reference's order (tiny_train.py:413-422)"""
    cfg = runpy.run_path(path)
    opt = {}
    for section in ("common", "model", "train", "optimizer"):
        opt.update(cfg.get(section, {}))
    return argparse.Namespace(**opt)


def make_learner(opt):
    return LEARNERS.get(opt.il, BaseLearner)(opt)


def synthetic_data(opt, classes_per_task=None):
    """device-generated crops + synthetic dictionaries standing in for the LMDB datasets (bench.py's data)"""
    from .data.synthetic import SyntheticTextLines, SyntheticValidation, synthetic_characters
    per_task = classes_per_task or [2086, 220, 1728, 1160, 73, 102][:len(opt.lan_list)]       # README.md:103 (MLT19)
    train, valid = SyntheticTextLines(opt, seed=opt.manual_seed + parallel.rank()), SyntheticValidation(opt)

    def characters(taski):
        chars = synthetic_characters(sum(per_task[:taski + 1]))
        train.set_characters(chars)
        valid.set_characters(chars)
        return chars
    return train, valid, characters


def train(opt, log, data=None):
    """the task loop of tiny_train.py:195-294.  data: optional (train_loader, valid_loader, characters(taski) -> str/list,
    test_loaders(taski) -> iterable of loaders) replacing the LMDB-backed managers (synthetic runs, tests)"""
    write_data_log(f"----------- {opt.exp_name} ------------\n")
    print(f"----------- {opt.exp_name} ------------\n")
    lan_list = list(opt.lan_list)
    best_scores, ned_scores, valid_datas, char = [], [], [], dict()
    learner = make_learner(opt)
    joint = opt.il in ("joint_loader", "joint_mix")
    if data is None:
        from .data.data_manage import Dataset_Manager, Val_Dataset
        from .data.dataset import AlignCollate
        data_manager = Dataset_Manager(opt, rank=parallel.rank(), world=parallel.world_size())      # every rank its own shard
        AlignCollate_valid = AlignCollate(opt, mode="test")
    else:
        data_manager, fixed_valid, characters, test_loaders = data
        AlignCollate_valid = None
    for taski in range(len(lan_list)):
        if data is None:
            for valid_data in opt.valid_datas:
                valid_datas.append(os.path.join(valid_data, lan_list[taski]))
            valid_loader = Val_Dataset(valid_datas, opt)
        else:
            valid_loader = fixed_valid
        if joint:                                              # joint training: one pass over the union of all tasks (:240-259)
            if data is None:
                valid_datas, char = [], {}
                for t in range(len(lan_list)):
                    for val_data in opt.valid_datas:
                        valid_datas.append(os.path.join(val_data, lan_list[t]))
                    data_manager.joint_start(opt, opt.select_data, log, t, len(lan_list))
                    for data_path in opt.select_data:
                        opt.character, char = load_dict(data_path + f"/{lan_list[t]}", char)
                tests = valid_datas
            else:
                opt.character = characters(len(lan_list) - 1)
                tests = list(test_loaders(len(lan_list) - 1))
            best_scores, ned_scores = learner.incremental_train(0, opt.character, data_manager, valid_loader, AlignCollate_valid, tests)
            best_scores, ned_scores = learner.test(AlignCollate_valid, tests, best_scores, ned_scores, 0)
            break
        if data is None:
            if taski == 0:
                data_manager.init_start(opt, opt.select_data, log, taski)
            for data_path in opt.select_data:                  # cumulative dictionary: the class set only grows (:265-269)
                opt.character, char = load_dict(data_path + f"/{lan_list[taski]}", char)
            tests = valid_datas
        else:
            if taski == 0:
                data_manager.init_start(opt, getattr(opt, "select_data", None), log, taski)
            opt.character = characters(taski)
            tests = list(test_loaders(taski))
        learner.incremental_train(taski, opt.character, data_manager, valid_loader)
        best_scores, ned_scores = learner.test(AlignCollate_valid, tests, best_scores, ned_scores, taski)
        learner.after_task()
    write_data_log(f"----------- {opt.exp_name} ------------\n")
    if best_scores:
        n_sets = len(getattr(opt, "valid_datas", [0]))
        if n_sets == 2:
            print("ALL Average 17 Acc: {:.2f} \n".format(sum(best_scores) / len(best_scores)))
            print("ALL Average 19 Acc: {:.2f} \n".format(sum(ned_scores) / len(ned_scores)))
            write_data_log("ALL 17 Acc: {:.2f} \n".format(sum(best_scores) / len(best_scores)))
            write_data_log("ALL 19 Acc: {:.2f} \n".format(sum(ned_scores) / len(ned_scores)))
        else:
            print("ALL Average Incremental Accuracy: {:.2f} \n".format(sum(best_scores) / len(best_scores)))
            write_data_log("ALL Average Acc: {:.2f} \n".format(sum(best_scores) / len(best_scores)))
    return learner, best_scores, ned_scores


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config/crnn_mrn.py", help="mmcv-style config file (the reference's config/*.py load as they are)")
    ap.add_argument("--synthetic", action="store_true", help="device-generated crops instead of the LMDB datasets")
    args = ap.parse_args(argv)
    opt = load_config(args.config)
    rank, world, local = parallel.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("mrn_amd.tiny_train needs an MI355X (there is no CPU fallback on the product path)")
    torch.cuda.set_device(local)
    torch.set_num_threads(max(1, host_cpu_budget() // max(1, world)))     # (PyTorch sizes its pools from the affinity mask, not the quota)
    seed_everything(opt.manual_seed)
    opt.gpu_name = "_".join(torch.cuda.get_device_name().split())
    opt.num_gpu = world
    if not getattr(opt, "exp_name", None):
        opt.exp_name = f"Seed{opt.manual_seed}-{opt.model_name}"
    os.makedirs(f"./saved_models/{opt.exp_name}", exist_ok=True)
    log = open(f"./saved_models/{opt.exp_name}/log_train.txt", "a")
    log.write("Command line input: python " + " ".join(sys.argv) + "\n")
    data = None
    if args.synthetic:
        train_loader, valid, characters = synthetic_data(opt)
        data = (train_loader, valid, characters, lambda taski: [valid.create_dataset()])
    train(opt, log, data=data)
    log.close()


if __name__ == "__main__":
    main()

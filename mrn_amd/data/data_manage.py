"""Dataset_Manager / Val_Dataset: per-task training loaders with rehearsal-memory mixing, and validation loaders
(reference data/data_manage.py:8-283).  Same duck-typed interface the learners consume --
`get_dataset(taski, memory=..., index_list=...)`, `get_batch()`, `get_batch2()`, `init_start`, `joint_start`,
`rehearsal_prev_model`, `Val_Dataset.create_dataset / create_list_dataset` -- and the same sampling semantics under the
same numpy / torch seeds (pinned against the reference's own classes by tests/golden/data_manage.npz).

MI355X-first addition: batches are staged to the GPU ahead of use -- collated into pinned host memory, copied on a side
HIP stream while the previous step computes, handed to the learner as device tensors (33.5 MB per 256-crop batch; the
reference's `image_tensors.to(device)` on pageable memory serialises behind the whole step).
"""
import bisect

import numpy.random
import torch
from torch.utils.data import ConcatDataset, DataLoader, Subset

from .dataset import AlignCollate, AlignCollate2, hierarchical_dataset, open_leaf

REPEAT_UP_TO = 50000          # "for faster training, we multiply small datasets itself" (data_manage.py:137-141)


class IndexConcatDataset(ConcatDataset):
    """ConcatDataset whose items carry the index of the member dataset they came from (reference :272-283): in MRN's router
    phase member 0 is the rehearsal memory of all previous tasks and member 1 the current task -- the router's domain label"""

    def __getitem__(self, idx):
        if idx < 0:
            if -idx > len(self):
                raise ValueError("absolute value of index should not exceed dataset length")
            idx = len(self) + idx
        dataset_idx = bisect.bisect_right(self.cumulative_sizes, idx)
        sample_idx = idx if dataset_idx == 0 else idx - self.cumulative_sizes[dataset_idx - 1]
        return self.datasets[dataset_idx][sample_idx], dataset_idx


class DeviceStager:
    """keeps ONE batch in flight to the GPU: pinned host copy -> async H2D on a side stream -> event the consumer stream waits on"""

    RING = 3        # pinned staging buffers: one being filled, one in flight, one whose copy the consumer may still wait on

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self._ring, self._turn = [], 0

    def _pinned(self, like):
        """a reusable pinned buffer of `like`'s shape whose previous copy has completed (allocating pinned memory per batch costs
        more than the whole copy: 70 of 78 ms per 256-crop batch measured with images.pin_memory())"""
        if len(self._ring) < self.RING:
            self._ring.append([torch.empty(like.shape, dtype=like.dtype, pin_memory=True), None])
            slot = self._ring[-1]
        else:
            slot = self._ring[self._turn % self.RING]
            self._turn += 1
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].shape != like.shape or slot[0].dtype != like.dtype:
                slot[0] = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
        return slot

    def upload(self, images):
        if self.stream is None:
            return images, None
        slot = self._pinned(images)
        slot[0].copy_(images)
        with torch.cuda.stream(self.stream):
            dev = slot[0].to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        slot[1] = done
        return dev, (done, slot[0])

    def ready(self, dev, ticket):
        if ticket is not None:
            torch.cuda.current_stream().wait_event(ticket[0])
            dev.record_stream(torch.cuda.current_stream())
        return dev


class Dataset_Manager(object):
    def __init__(self, opt, open_dataset=None, device=None, rank=0, world=1):
        """open_dataset(path, opt, mode) -> Dataset of (PIL RGBA image, label): defaults to the LMDB / NPZ leaf reader; tests and
        array-backed pipelines pass their own.  device: where batches are staged (None: cuda when available, else no staging).
        rank / world: data parallelism (one process per GPU, mrn_amd/parallel.py).  torch.nn.DataParallel scatters ONE loader's batch of
        opt.batch_size over the GPUs (reference il_modules/base.py:68); here every rank owns its loaders, so with world > 1 each rank
        draws batch_size // world samples per loader from its OWN shuffle order (a torch.Generator seeded manual_seed + rank): the
        ranks see different samples and the global batch stays opt.batch_size.  The numpy seed stays common, so the rehearsal-memory
        index sets agree between the ranks.  world == 1 leaves everything as it was (global RNG, full batch)."""
        self.rank, self.world = int(rank), max(1, int(world))
        self.data_list = []
        self.data_loader_list = []
        self.dataloader_iter_list = []
        self.select_data = None
        self.opt = opt
        self.open_dataset = open_dataset or open_leaf
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.stager = DeviceStager(device) if getattr(opt, "device_prefetch", True) else DeviceStager(torch.device("cpu"))
        self._staged = None

    # -- the loaders of one task (reference :16-61) ------------------------------------------------------------------
    def get_dataset(self, taski, memory="random", index_list=None):
        self.data_loader_list = []
        self.dataloader_iter_list = []
        self._staged = None
        memory_num = self.opt.memory_num
        dataset = self.create_dataset(data_list=self.select_data, taski=taski)
        if memory is not None and self.opt.il == "mrn":
            # current task: memory_num / taski random samples; memory: memory_num / taski samples of every previous task
            index_current = numpy.random.choice(range(len(dataset)), int(memory_num / taski), replace=False)
            split_dataset = Subset(dataset, index_current.tolist())
            memory_data, index_list = self.rehearsal_memory(taski, random=False, total_num=memory_num, index_array=index_list)
            self.create_dataloader_mix(IndexConcatDataset([memory_data, split_dataset]), self.opt.batch_size)
            print("taski is {} current dataset chose {}\n now dataset chose {}".format(taski, int(memory_num / taski), len(memory_data)))
        elif memory == "test_ch":
            memory_data, index_list = self.rehearsal_memory(taski, random=False, total_num=memory_num, index_array=index_list, repeat=True)
            self.create_dataloader_mix(IndexConcatDataset([memory_data, dataset]), self.opt.batch_size)
        elif memory == "large":
            index_current = numpy.random.choice(range(len(dataset)), memory_num, replace=False)
            split_dataset = Subset(dataset, index_current.tolist())
            memory_data, index_list = self.rehearsal_memory(taski, random=False, total_num=memory_num * taski, index_array=index_list)
            self.create_dataloader_mix(IndexConcatDataset([memory_data, split_dataset]), self.opt.batch_size)
        elif memory == "total":
            total = [dataset] + [self.create_dataset(data_list=self.select_data, taski=i) for i in range(taski)]
            self.create_dataloader_mix(IndexConcatDataset(total), self.opt.batch_size)
        elif memory is not None:
            # two balanced half-batches: rehearsal memory and current task
            memory_data, index_list = self.rehearsal_memory(taski, random=False, total_num=memory_num, index_array=index_list)
            self.create_dataloader(memory_data, self.opt.batch_size // 2)
            self.create_dataloader(dataset, self.opt.batch_size // 2)
        else:
            self.create_dataloader(dataset)
        return index_list

    def joint_start(self, opt, select_data, log, taski, total_task):
        self.opt = opt
        self.select_data = select_data
        log.write("-" * 80 + "\n")
        dataset = self.create_dataset(data_list=self.select_data, taski=taski)
        if opt.il == "joint_mix":
            self.data_list.append(dataset)
            if taski == total_task - 1:
                self.create_dataloader(ConcatDataset(self.data_list), int(self.opt.batch_size))
        elif opt.il == "joint_loader":
            self.create_dataloader(dataset, int(self.opt.batch_size // total_task))

    def init_start(self, opt, select_data, log, taski):
        self.opt = opt
        self.select_data = select_data
        self.data_loader_list = []
        self.dataloader_iter_list = []
        print(f"select_data: {select_data}\n")
        log.write("-" * 80 + "\n" + f"select_data: {select_data}\n")
        self.get_dataset(taski, memory=None)

    def rehearsal_memory(self, taski, random=False, total_num=2000, index_array=None, repeat=False):
        """Subset of every previous task's dataset: the learner's index arrays (il_modules/base.py:278-302), or fresh random ones"""
        data_list = []
        num_i = int(total_num / taski)
        print("memory size is {}\n".format(num_i))
        for i in range(taski):
            dataset = self.create_dataset(data_list=self.select_data, taski=i, repeat=repeat)
            index_list = numpy.random.choice(range(len(dataset)), num_i, replace=repeat) if random else index_array[i]
            data_list.append(Subset(dataset, index_list.tolist()))
        return ConcatDataset(data_list), index_array

    def rehearsal_prev_model(self, taski):
        dataset = self.create_dataset(data_list=self.select_data, taski=taski - 1, repeat=False)
        data_loader = DataLoader(dataset, batch_size=self.opt.batch_size, shuffle=False, num_workers=int(self.opt.workers),
                                 collate_fn=AlignCollate(self.opt), pin_memory=False, drop_last=False)
        return data_loader, len(dataset)

    def create_dataset(self, data_list="/", taski=0, mode="train", repeat=True):
        """one dataset per root in data_list (<root>/<language of task taski>), small ones repeated up to 50 000 samples"""
        dataset_list = []
        for data_root in data_list:
            dataset = self.open_dataset(data_root + "/" + self.opt.lan_list[taski], self.opt, mode)
            print(f"num samples: {len(dataset)}")
            if len(dataset) < REPEAT_UP_TO and repeat:
                dataset = ConcatDataset([dataset] * int(REPEAT_UP_TO / len(dataset)))
            dataset_list.append(dataset)
        return ConcatDataset(dataset_list)

    def _loader(self, dataset, batch_size, collate):
        bs = self.opt.batch_size if batch_size is None else batch_size
        gen = None
        if self.world > 1:
            if bs % self.world:          # the global batch silently changes otherwise (truncation; max(1, ..) can even inflate it)
                print(f"[Dataset_Manager] per-loader batch {bs} is not divisible by the {self.world} ranks: every rank draws "
                      f"{max(1, bs // self.world)}, global batch {max(1, bs // self.world) * self.world} instead of {bs}")
            bs = max(1, bs // self.world)
            gen = torch.Generator()
            gen.manual_seed(int(getattr(self.opt, "manual_seed", 0)) + 7919 * (self.rank + 1) + 104729 * len(self.data_loader_list))
        loader = DataLoader(dataset, batch_size=bs, shuffle=True, generator=gen,
                            num_workers=int(self.opt.workers), collate_fn=collate, pin_memory=False, drop_last=False)
        self.data_loader_list.append(loader)
        self.dataloader_iter_list.append(iter(loader))

    def create_dataloader(self, dataset, batch_size=None):
        self._loader(dataset, batch_size, AlignCollate(self.opt))

    def create_dataloader_mix(self, dataset, batch_size=None):
        self._loader(dataset, batch_size, AlignCollate2(self.opt))

    # -- batches (reference :174-217) ----------------------------------------------------------------------------------
    def _next(self, i):
        try:
            return next(self.dataloader_iter_list[i])
        except StopIteration:
            self.dataloader_iter_list[i] = iter(self.data_loader_list[i])
            return next(self.dataloader_iter_list[i])

    def _host_batch(self, with_index):
        images, labels, index = [], [], []
        for i in range(len(self.dataloader_iter_list)):
            try:
                got = self._next(i)
            except ValueError:
                continue
            images.append(got[0])
            labels += got[1]
            if with_index:
                index.append(got[2])
        return (images[0] if len(images) == 1 else torch.cat(images, 0)), labels, index

    def _staged_batch(self, with_index):
        """the batch prepared by the previous call (already on its way to the GPU) + start the next one"""
        key = "mix" if with_index else "plain"
        if self._staged is None or self._staged[0] != key:
            images, labels, index = self._host_batch(with_index)
            self._staged = (key, self.stager.upload(images), labels, index)
        _, (dev, ticket), labels, index = self._staged
        nxt = self._host_batch(with_index)
        self._staged = (key, self.stager.upload(nxt[0]), nxt[1], nxt[2])
        return self.stager.ready(dev, ticket), labels, index

    def get_batch(self):
        if self.stager.stream is None:
            images, labels, _ = self._host_batch(False)
            return images, labels
        images, labels, _ = self._staged_batch(False)
        return images, labels

    def get_batch2(self):
        if self.stager.stream is None:
            return self._host_batch(True)
        return self._staged_batch(True)


class Val_Dataset(object):
    def __init__(self, val_datas, opt, open_tree=None):
        """open_tree(root, opt, mode) -> (dataset, log): defaults to hierarchical_dataset over LMDB / NPZ leaves"""
        self.data_loader_list = []
        self.dataset_list = []
        self.current_data = val_datas[-1]
        self.val_datas = val_datas
        self.opt = opt
        self.AlignCollate_valid = AlignCollate(self.opt, mode="test")
        self.open_tree = open_tree or (lambda root, opt, mode: hierarchical_dataset(root=root, opt=opt, mode=mode))

    def _loader(self, dataset):
        return DataLoader(dataset, batch_size=self.opt.batch_size, shuffle=True, num_workers=int(self.opt.workers),
                          collate_fn=self.AlignCollate_valid, pin_memory=False)

    def create_dataset(self, val_data=None):
        valid_dataset, _ = self.open_tree(self.current_data if val_data is None else val_data, self.opt, "test")
        print("-" * 80)
        return self._loader(valid_dataset)

    def create_list_dataset(self, valid_datas=None):
        """every validation set seen so far, at most 700 random samples of each (reference :247-269)"""
        concat = []
        for val_data in (self.val_datas if valid_datas is None else valid_datas):
            valid_dataset, log = self.open_tree(val_data, self.opt, "test")
            if len(valid_dataset) > 700:
                index_current = numpy.random.choice(range(len(valid_dataset)), 700, replace=False)
                valid_dataset = Subset(valid_dataset, index_current.tolist())
            concat.append(valid_dataset)
            print(log)
            print("-" * 80)
        return self._loader(ConcatDataset(concat))


def evaluation_loader(root, opt, collate=None):
    """the per-test-set loader of BaseLearner.test (il_modules/base.py:379-389)"""
    dataset, _ = hierarchical_dataset(root=root, opt=opt, mode="test")
    return DataLoader(dataset, batch_size=opt.batch_size, shuffle=True, num_workers=int(opt.workers),
                      collate_fn=collate or AlignCollate(opt, mode="test"), pin_memory=False)

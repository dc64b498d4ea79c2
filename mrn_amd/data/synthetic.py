"""Synthetic stand-in for the reference's Dataset_Manager / Val_Dataset (data/data_manage.py:8-283), keeping the
duck-typed interface the learners consume: get_batch() -> (FloatTensor[B,4,32,256], list[str]),
get_batch2() -> (..., ..., [tuple[int]*B]), get_dataset(...), init_start(...), rehearsal_prev_model(...).
The LMDB reader, augmentation and rehearsal-memory sampling are out of scope."""
import numpy as np
import torch


class SyntheticTextLines:
    """U(-1,1) RGBA 32x256 crops (generated on the device) with random labels over a character set."""

    def __init__(self, opt, device=None, seed=111):
        self.opt = opt
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.rng = np.random.default_rng(seed)
        self.characters = ""
        self.taski = 0

    # -- reference interface ---------------------------------------------------------------------------
    def init_start(self, *args, **kwargs):
        """Dataset_Manager.init_start(opt, select_data, log, taski) (data/data_manage.py:80-95); init_start(taski) also accepted"""
        self.taski = kwargs.get("taski", args[-1] if args else 0)

    def get_dataset(self, taski, memory=None, index_list=None):
        """same contract as Dataset_Manager.get_dataset (data/data_manage.py:16-61): with a rehearsal memory the learner must
        hand over one index array per previous task (rehearsal_memory() indexes index_array[i] for i < taski)"""
        self.taski = taski
        if memory is not None and getattr(self.opt, "il", None) == "mrn":
            assert index_list is not None and len(index_list) >= taski, "rehearsal memory needs one index array per previous task"
            for idx in index_list[:taski]:
                assert len(idx) > 0 and int(max(idx)) < self.dataset_len
        return index_list

    dataset_len = 50000          # nominal samples per task (the reference repeats small datasets up to 50k, data_manage.py:137-141)

    def rehearsal_prev_model(self, taski):
        return self, self.dataset_len

    def set_characters(self, characters):
        self.characters = characters

    def _labels(self, B):
        n = max(len(self.characters), 1)
        lens = self.rng.integers(1, self.opt.batch_max_length + 1, size=B)
        out = []
        for L in lens:
            ids = self.rng.integers(0, n, size=int(L))
            out.append("".join(self.characters[i] for i in ids) if self.characters else "")
        return out

    def _images(self, B):
        shape = (B, self.opt.input_channel, self.opt.imgH, self.opt.imgW)
        return torch.rand(shape, generator=self.gen, device=self.device) * 2 - 1

    def get_batch(self):
        B = self.opt.batch_size
        return self._images(B), self._labels(B)

    def get_batch2(self):
        B = self.opt.batch_size
        index = [tuple(int(v) for v in self.rng.integers(0, 2, size=B))]     # 0 = rehearsal memory, 1 = current task
        return self._images(B), self._labels(B), index


class SyntheticValidation:
    """Val_Dataset stand-in: create_dataset()/create_list_dataset() return an iterable of (images, labels)."""

    def __init__(self, opt, batches=1, device=None, seed=7):
        self.src = SyntheticTextLines(opt, device, seed)
        self.batches = batches

    def set_characters(self, characters):
        self.src.set_characters(characters)

    def create_dataset(self):
        return [self.src.get_batch() for _ in range(self.batches)]

    def create_list_dataset(self):
        return self.create_dataset()


def synthetic_characters(n, start=0x4E00):
    """a character set of n distinct code points (CJK block) standing in for a language dictionary"""
    return "".join(chr(start + i) for i in range(n))

"""Datasets and collation of the input pipeline (reference data/dataset.py:16-246).

Same on-disk format, classes, constructor arguments and tensor contract as the reference -- `[B, 4, 32, 256]` fp32 in
[-1, 1] from RGBA crops resized BICUBIC -- so that the reference's datasets (tools/create_lmdb_dataset.py: LMDB keys
`num-samples`, `label-%09d`, `image-%09d`) feed this package unchanged:

  * LmdbDataset      -- `lmdb.open` when the package is installed, otherwise the read-only data.mdb walker of mrn_amd/data/mdb.py
                        (the package is absent from this image);
  * ArrayDataset     -- the same samples from memory (lists / arrays of RGBA crops): what the tests and array-backed
                        pipelines use, and the in-memory stand-in when LMDB is not installed;
  * NpzDataset       -- a directory holding `data.npz` (`images`: object array of encoded image bytes or uint8 arrays,
                        `labels`): the portable sibling of an LMDB leaf directory;
  * ResizeNormalize / AlignCollate / AlignCollate2 / hierarchical_dataset -- as in the reference (no torchvision needed).

The reference's augmentation classes (cv2-based, data/transform.py) are out of scope: every shipped config sets Aug="None".
"""
import io
import os
import sys

import numpy as np
import PIL.Image
import torch
from torch.utils.data import ConcatDataset, Dataset


def _open_rgba(buf_or_array, opt):
    """decode one stored sample to an RGBA PIL image; corrupted images become a blank crop with a dummy label marker"""
    if isinstance(buf_or_array, PIL.Image.Image):
        return buf_or_array.convert("RGBA"), True
    if isinstance(buf_or_array, np.ndarray) and buf_or_array.dtype == np.uint8 and buf_or_array.ndim == 3:
        return PIL.Image.fromarray(buf_or_array).convert("RGBA"), True
    try:
        data = bytes(buf_or_array) if not isinstance(buf_or_array, (bytes, bytearray)) else buf_or_array
        return PIL.Image.open(io.BytesIO(data)).convert("RGBA"), True
    except (IOError, OSError, ValueError):
        return PIL.Image.new("RGBA", (opt.imgW, opt.imgH)), False


class _LabelledImages(Dataset):
    """shared behaviour: labels longer than opt.batch_max_length are filtered out at construction (dataset.py:78-83)"""

    def _filter(self, labels, opt):
        self.filtered_index_list = [i for i, lab in enumerate(labels) if lab is not None and len(lab) <= opt.batch_max_length]
        self.nSamples = len(self.filtered_index_list)

    def __len__(self):
        return self.nSamples


class ArrayDataset(_LabelledImages):
    """samples from memory: images = sequence of PIL images / uint8 [H,W,C] arrays / encoded bytes, labels = sequence of str"""

    def __init__(self, images, labels, opt, mode="train"):
        assert len(images) == len(labels), "Data size error!"
        self.images, self.labels, self.opt, self.mode = images, list(labels), opt, mode
        self._filter(self.labels, opt)

    def __getitem__(self, index):
        assert index <= len(self), "index range error"
        i = self.filtered_index_list[index]
        img, ok = _open_rgba(self.images[i], self.opt)
        return (img, self.labels[i] if ok else "[dummy_label]")


class NpzDataset(ArrayDataset):
    """<root>/data.npz with `images` (object array: encoded bytes or uint8 arrays) and `labels` (array of str)"""

    def __init__(self, root, opt, mode="train"):
        z = np.load(os.path.join(root, "data.npz"), allow_pickle=True)
        super().__init__(list(z["images"]), [str(s) for s in z["labels"]], opt, mode)
        self.root = root


class LmdbDataset(_LabelledImages):
    """reference data/dataset.py:44-112: LMDB environment with keys num-samples / label-%09d / image-%09d (1-based)"""

    def __init__(self, root, opt, mode="train"):
        from .mdb import open_environment
        self.root, self.opt, self.mode = root, opt, mode
        # the `lmdb` package when installed (lmdb.open(root, readonly=True, lock=False, ...) as in the reference), otherwise the
        # read-only B+tree walker of mrn_amd/data/mdb.py over the same data.mdb
        self.env = open_environment(root)
        if not self.env:
            print("cannot open lmdb from %s" % root)
            sys.exit(0)
        with self.env.begin(write=False) as txn:
            n = int(txn.get("num-samples".encode()))
            labels = []
            for index in range(1, n + 1):
                raw = txn.get("label-%09d".encode() % index)
                labels.append(None if raw is None else raw.decode("utf-8"))
        self._filter(labels, opt)
        self.filtered_index_list = [i + 1 for i in self.filtered_index_list]       # lmdb indices start at 1

    def __getitem__(self, index):
        assert index <= len(self), "index range error"
        index = self.filtered_index_list[index]
        with self.env.begin(write=False) as txn:
            label = txn.get("label-%09d".encode() % index).decode("utf-8")
            img, ok = _open_rgba(txn.get("image-%09d".encode() % index), self.opt)
        return (img, label if ok else "[dummy_label]")


def open_leaf(dirpath, opt, mode="train"):
    """one leaf directory of a dataset tree: LMDB (data.mdb) or its portable sibling (data.npz)"""
    if os.path.exists(os.path.join(dirpath, "data.npz")):
        return NpzDataset(dirpath, opt, mode=mode)
    return LmdbDataset(dirpath, opt, mode=mode)


def hierarchical_dataset(root, opt, select_data="/", data_type="label", mode="train"):
    """every leaf directory under root whose path contains one of select_data (reference :16-41) -> (ConcatDataset, log)"""
    dataset_list = []
    dataset_log = f"dataset_root:  {root}\t dataset: {select_data}"
    print(dataset_log)
    dataset_log += "\n"
    for dirpath, dirnames, filenames in os.walk(root + "/"):
        if not dirnames:
            if any(sel in dirpath for sel in select_data):
                dataset = open_leaf(dirpath, opt, mode=mode)
                sub = f"sub-directory:\t/{os.path.relpath(dirpath, root)}\t num samples: {len(dataset)}"
                print(sub)
                dataset_log += f"{sub}\n"
                dataset_list.append(dataset)
    return ConcatDataset(dataset_list), dataset_log


class ResizeNormalize(object):
    """PIL resize to (W, H) + ToTensor + (x - 0.5) / 0.5 (reference :235-246): [C, H, W] fp32 in [-1, 1]"""

    def __init__(self, size, interpolation=PIL.Image.BICUBIC):
        self.size = size                 # CAUTION: (width, height), as PIL wants it
        self.interpolation = interpolation

    def __call__(self, image):
        image = image.resize(self.size, self.interpolation)
        a = np.asarray(image, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div_(255.0)      # ToTensor
        return t.sub_(0.5).div_(0.5)


class AlignCollate(object):
    """batch of (PIL image, label) -> (FloatTensor [B,C,H,W], tuple of labels) (reference :171-197)"""

    def __init__(self, opt, mode="train"):
        self.opt, self.mode = opt, mode
        if getattr(opt, "Aug", "None") == "None" or mode != "train":
            self.transform = ResizeNormalize((opt.imgW, opt.imgH))
        else:
            raise NotImplementedError(f"Aug='{opt.Aug}': the cv2-based augmentations are out of scope (every shipped config uses 'None')")

    def __call__(self, batch):
        images, labels = zip(*batch)
        return torch.stack([self.transform(image) for image in images], 0), labels


class AlignCollate2(AlignCollate):
    """batch of ((PIL image, label), dataset index) from IndexConcatDataset -> (images, labels, index) (reference :144-170)"""

    def __call__(self, batch):
        b_info, index = zip(*batch)
        images, labels = zip(*b_info)
        return torch.stack([self.transform(image) for image in images], 0), labels, index

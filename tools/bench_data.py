#!/usr/bin/env python3
"""Throughput of the real-data input pipeline: Dataset_Manager over an in-memory ArrayDataset of encoded PNG crops
(decode -> RGBA -> BICUBIC resize to 32x256 -> [-1,1], DataLoader workers, pinned staging + side-stream H2D) -- batches per second
handed to the learner, for several worker counts.  On a GPU box the batches arrive as device tensors.
    python tools/bench_data.py [samples] [batches]"""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np
import PIL.Image
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd.data.data_manage import Dataset_Manager  # noqa: E402
from mrn_amd.data.dataset import ArrayDataset  # noqa: E402
from mrn_amd.tools.utils import host_cpu_budget  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rng = np.random.default_rng(0)
    images, labels = [], []
    for i in range(n):                                   # word crops of varying width, PNG-encoded like an LMDB value
        w = int(rng.integers(40, 200))
        arr = rng.integers(0, 255, (int(rng.integers(24, 48)), w, 3), dtype=np.uint8)
        buf = io.BytesIO()
        PIL.Image.fromarray(arr).save(buf, format="PNG")
        images.append(buf.getvalue())
        labels.append("".join(chr(97 + int(c)) for c in rng.integers(0, 26, int(rng.integers(1, 20)))))
    cores = host_cpu_budget()
    torch.set_num_threads(cores)          # as the driver does (mrn_amd/tiny_train.py)
    print(f"{n} PNG crops in memory, batch 256, {cores} host cores, cuda={torch.cuda.is_available()}")
    for workers in [w for w in (0, 4, 8, 12, 16, 32) if w <= cores]:
        opt = argparse.Namespace(imgH=32, imgW=256, batch_max_length=25, batch_size=256, workers=workers, lan_list=["x"], il="base",
                                 memory_num=2000, PAD=False, Aug="None", data_filtering_off=True, select_data=["mem"])
        with contextlib.redirect_stdout(io.StringIO()):
            dm = Dataset_Manager(opt, open_dataset=lambda path, o, mode: ArrayDataset(images, labels, o, mode))
            dm.init_start(opt, ["mem"], io.StringIO(), 0)
        for _ in range(3):
            img, lab = dm.get_batch()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(batches):
            img, lab = dm.get_batch()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"workers {workers:2d}: {batches * 256 / dt:8.0f} images/s  ({dt / batches * 1e3:6.1f} ms per 256-crop batch, "
              f"batch on {img.device}, {tuple(img.shape)})", flush=True)
        del dm


if __name__ == "__main__":
    main()

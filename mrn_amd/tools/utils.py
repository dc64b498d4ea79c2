"""Label converters and the loss averager (reference tools/utils.py:10-166) -- the integer label path.
Same class names, constructor arguments, dict layouts and return types; tensors go to the current device."""
import numpy as np
import torch

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def to_device(t):
    """host tensor -> current device without stalling the host: a pageable H2D copy waits for everything queued on the
    stream, which would serialise the label encoding of step n+1 behind the GPU work of step n"""
    if device.type != "cuda":
        return t
    return t.pin_memory().to(device, non_blocking=True)


def _padded_rows(rows, width, pad):
    """list of index lists -> long tensor [len(rows), width], short rows filled with `pad` (one array fill instead of a tensor
    construction and a slice assignment per word: 1.5 ms -> 0.2 ms for 256 labels, host time of a launch-bound step); a row longer
    than `width` is an error, as the slice assignment of the reference's encode is (tools/utils.py:52-58,122-131)"""
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    if len(rows) and int(lens.max()) > width:
        raise RuntimeError(f"a label of {int(lens.max())} tokens does not fit the {width} columns of the batch")
    out = np.full((len(rows), width), pad, dtype=np.int64)
    out[np.arange(width)[None, :] < lens[:, None]] = np.fromiter((v for r in rows for v in r), dtype=np.int64, count=int(lens.sum()))
    return torch.from_numpy(out)


class CTCLabelConverter(object):
    """index 0 = CTC blank; then [PAD]=1, [UNK]=2, ' '=3, then the characters in order."""

    def __init__(self, character):
        tokens = ["[PAD]", "[UNK]", " "] + list(character)
        self.dict = {}
        for pos, ch in enumerate(tokens, start=1):      # a repeated character keeps its LAST index
            self.dict[ch] = pos
        self.character = ["[CTCblank]"] + tokens
        print(f"# characters dict has: {len(self.character)}")

    def encode(self, word_string, batch_max_length=25):
        lengths = [len(w) for w in word_string]
        pad, unk = self.dict["[PAD]"], self.dict["[UNK]"]
        index = _padded_rows([[self.dict.get(ch, unk) for ch in word] for word in word_string], batch_max_length, pad)
        return to_device(index), to_device(torch.IntTensor(lengths))

    def decode(self, word_index, word_length):
        """greedy CTC collapse: drop blanks (0) and merge repeats"""
        texts = []
        for row, n in zip(word_index, word_length):
            prev, chars = None, []
            for k in list(row[:int(n)]):
                k = int(k)
                if k != 0 and k != prev:
                    chars.append(self.character[k])
                prev = k
            texts.append("".join(chars))
        return texts


class AttnLabelConverter(object):
    """[UNK]=0 [PAD]=1 [SOS]=2 [EOS]=3 ' '=4, then the characters."""

    def __init__(self, character):
        self.character = ["[UNK]", "[PAD]", "[SOS]", "[EOS]", " "] + list(character)
        self.dict = {}
        for pos, ch in enumerate(self.character):
            self.dict[ch] = pos
        print(f"# of tokens and characters: {len(self.character)}")

    def encode(self, word_string, batch_max_length=25):
        lengths = [len(w) + 1 for w in word_string]          # + [EOS]
        width = batch_max_length + 2                         # [SOS] + text + [EOS]
        unk, sos, eos = self.dict["[UNK]"], self.dict["[SOS]"], self.dict["[EOS]"]
        index = _padded_rows([[sos] + [self.dict.get(ch, unk) for ch in word] + [eos] for word in word_string], width, self.dict["[PAD]"])
        return to_device(index), to_device(torch.IntTensor(lengths))

    def decode(self, word_index, word_length):
        return ["".join(self.character[int(k)] for k in row[:int(n)]) for row, n in zip(word_index, word_length)]


class Averager(object):
    """running mean of loss tensors"""

    def __init__(self):
        self.reset()

    def add(self, v):
        self.n_count += v.data.numel()
        self.sum += v.data.sum()

    def reset(self):
        self.n_count = 0
        self.sum = 0

    def val(self):
        return self.sum / float(self.n_count) if self.n_count != 0 else 0


def adjust_learning_rate(optimizer, iteration, opt):
    """stepwise decay used when opt.schedule is a list of milestones (fractions of num_iter)"""
    lr = opt.lr
    for milestone in opt.schedule:
        if iteration >= float(milestone) * opt.num_iter:
            lr *= opt.lr_drop_rate
    for group in optimizer.param_groups:
        group["lr"] = lr


def host_cpu_budget():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (containers often expose every core of
    the host in the mask -- 256 on the MI355X boxes -- while the quota allows 16; PyTorch sizes its thread pools from the mask and
    oversubscribes 16x, which made a 33 MB torch.cat take 40 ms)"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                 # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:                                                      # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n

"""Shared helpers for the parity tests (CPU and GPU)."""
import os

import numpy as np
import torch

from mrn_amd.tools import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_state_dict(g, seed):
    """Rebuild the reference model's state_dict (key layout from the fixture, values from the generator)."""
    sd = {}
    for k, shp in zip(g["sd_keys"], g["sd_shapes"]):
        k = str(k)
        shape = tuple(int(v) for v in str(shp).split(",")) if str(shp) else ()
        sd[k] = torch.from_numpy(np.array(W.det_param(W.canonical_key(k), shape, seed)))
    return sd


def sub(t, n=4096):
    a = t.detach().cpu().double().numpy().reshape(-1)
    step = max(1, a.size // n)
    return a[::step][:n].astype(np.float32), a.mean(), np.abs(a).mean()


def assert_sub_close(g, name, t, atol=1e-4, rtol=1e-4):
    """Compare a tensor with a stored strided subsample + moments."""
    assert tuple(g[name + "/shape"]) == tuple(t.shape), (name, tuple(g[name + "/shape"]), tuple(t.shape))
    s, mean, absmean = sub(t)
    ref = g[name + "/sub"]
    err = np.abs(s - ref).max()
    tol = atol + rtol * np.abs(ref).max()
    assert err <= tol, f"{name}: max abs err {err:.3e} > tol {tol:.3e}"
    assert abs(mean - float(g[name + "/mean"])) <= atol + rtol * abs(float(g[name + "/absmean"])), name
    return err


def assert_close(name, a, b, atol=1e-4, rtol=1e-4):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    tol = atol + rtol * (np.abs(b).max() if b.size else 0.0)
    assert err <= tol, f"{name}: max abs err {err:.3e} > tol {tol:.3e}"
    return err


def det_inputs(kind, classes, B, seed, noise=False):
    """Same synthetic batch the golden generator used (tests/golden/make_golden.py: run())."""
    if noise:
        image = torch.from_numpy(W.uniform("input:image", (B, 4, 32, 256), -1.0, 1.0, seed))
    else:
        image = torch.from_numpy(W.smooth_image("input:image", (B, 4, 32, 256), seed))
    nspecial = 4 if kind in ("crnn", "svtr") else 5
    nchar = classes[-1] - nspecial
    chars = "".join(chr(0x4E00 + i) for i in range(nchar))
    lens = W.randint("label_len", (B,), 1, 26, seed)
    words = []
    for b in range(B):
        ids = W.randint(f"label_{b}", (int(lens[b]),), 0, nchar, seed)
        words.append("".join(chars[i] for i in ids))
    domain = torch.from_numpy(W.randint("domain", (B,), 0, 2, seed))
    return image, words, chars, domain


def assert_sub_l2(g, name, t, rel=0.05, q=0.99, q_atol=6e-6):
    """Robust comparison for Adam parameter deltas: elements whose gradient is ~0 get a normalised update m/sqrt(v)
    of essentially random sign, so a max-abs check is meaningless there.  Require a small relative L2 error over the
    stored subsample and a tight bound on the q-quantile of the absolute error."""
    assert tuple(g[name + "/shape"]) == tuple(t.shape), name
    s, _, _ = sub(t)
    ref = g[name + "/sub"]
    err = np.abs(s - ref)
    l2 = np.linalg.norm(s - ref) / max(np.linalg.norm(ref), 1e-30)
    assert l2 <= rel, f"{name}: relative L2 error {l2:.3e} > {rel}"
    assert np.quantile(err, q) <= q_atol, f"{name}: {q}-quantile abs err {np.quantile(err, q):.3e} > {q_atol}"
    return l2


def drop_masks(B, seed, tag, n_experts=1):
    """the DropPath draws the golden generator injected into the reference (tests/golden/make_golden.py)"""
    return [[torch.from_numpy(W.randint(f"droppath:{tag}:{e}:{k}", (B,), 0, 2, seed)).float() for k in range(22)]
            for e in range(n_experts)]


class DetLoader:
    """Deterministic stand-in for the reference's Dataset_Manager / Val_Dataset (data/data_manage.py:8-283): every batch is
    a pure function of (tag, seed, call number), so the golden generator (driving the REFERENCE learners) and the tests
    (driving the HIP learners) see byte-identical batches.  Labels are drawn over the characters set by set_characters();
    `oov` appends a character outside every dictionary to the first label of each batch (the [UNK] path)."""

    def __init__(self, B, tag, seed, noise=False, oov=False, n_valid=1):
        self.B, self.tag, self.seed, self.noise, self.oov, self.n_valid = B, tag, seed, noise, oov, n_valid
        self.chars = ""
        self.count = 0
        self.calls = []            # (method, args) log: lets tests assert how a learner drove the loader

    def set_characters(self, chars):
        self.chars = chars

    def _batch(self, n):
        name = f"{self.tag}:{n}"
        if self.noise:
            image = W.uniform(name + ":img", (self.B, 4, 32, 256), -1.0, 1.0, self.seed)
        else:
            image = W.smooth_image(name + ":img", (self.B, 4, 32, 256), self.seed)
        lens = W.randint(name + ":len", (self.B,), 1, 26, self.seed)
        labels = []
        for b in range(self.B):
            ids = W.randint(f"{name}:lab{b}", (int(lens[b]),), 0, len(self.chars), self.seed)
            labels.append("".join(self.chars[i] for i in ids))
        if self.oov:
            labels[0] = (labels[0][:24] + "é")
        return torch.from_numpy(image), labels

    # -- Dataset_Manager interface -------------------------------------------------------------------------
    def init_start(self, *a, **k):
        self.calls.append(("init_start", a))

    def get_dataset(self, taski, memory=None, index_list=None):
        self.calls.append(("get_dataset", (taski, memory, None if index_list is None else [np.asarray(i).copy() for i in index_list])))
        return index_list

    def rehearsal_prev_model(self, taski):
        self.calls.append(("rehearsal_prev_model", (taski,)))
        return self, 500

    def get_batch(self):
        self.count += 1
        return self._batch(self.count - 1)

    def get_batch2(self):
        image, labels = self.get_batch()
        index = tuple(int(v) for v in W.randint(f"{self.tag}:{self.count - 1}:dom", (self.B,), 0, 2, self.seed))
        return image, labels, [index]

    # -- Val_Dataset interface -----------------------------------------------------------------------------
    def create_dataset(self, val_data=None):
        return [self._batch(10_000 + i) for i in range(self.n_valid)]

    def create_list_dataset(self, valid_datas=None):
        return self.create_dataset()


class oracle_dtype:
    """run the CPU oracle in another floating-point precision (float64: the exact-arithmetic yardstick for conditioning
    bands): sets torch's default dtype and casts the oracle's TPS constants"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        from oracle import mrn_oracle as O
        self.O, self.old = O, O.tps_constants
        dt = self.dtype
        O.tps_constants = lambda *a: tuple(t.to(dt) for t in self.old(*a))
        torch.set_default_dtype(dt)
        return self

    def __exit__(self, *exc):
        torch.set_default_dtype(torch.float32)
        self.O.tps_constants = self.old

    def cast(self, sd):
        return {k: (v.to(self.dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}


def crafted_validation_case(kind):
    """Two evaluation batches with hand-made logits for validation() (reference test.py:139-279): every scoring branch is hit --
    exact match, a wrong character (fractional normalised edit distance), an out-of-dictionary label character ([UNK] never
    counts as correct), an empty prediction, an empty ground truth and, for the attention head, predictions with an [EOS]
    (pruned there) and without one (the reference then drops the LAST character: prd[:prd.find('[EOS]')] with find = -1).
    Returns (chars, batches [(image, labels)], logits [tensor [B,T,C]]): the stub model returns logits[i] for batch i."""
    chars = "".join(chr(0x4E00 + i) for i in range(36))
    ctc = kind == "crnn"
    T, C = (63, 40) if ctc else (26, 41)
    first = 4 if ctc else 5                                  # index of chars[0] in the converters' tables

    def ids(word):
        return [first + chars.index(c) for c in word]

    def attn_row(tokens):                                    # tokens then [EOS]=3 ... ; None = no [EOS] anywhere
        return (tokens + [3] + [first] * T)[:T]

    def ctc_row(tokens):                                     # every character twice, blanks between characters, then blanks
        seq = []
        for t in tokens:
            seq += [t, t, 0]
        return (seq + [0] * T)[:T]

    w = [chars[:5], chars[3:12], chars[7:9], chars[10:14] + "é", chars[20:23], ""]
    batches, logits = [], []
    for bi in range(2):
        labels = [w[(i + 3 * bi) % len(w)] for i in range(3)] if bi == 0 else [w[3], w[4], w[5]]
        rows = []
        for si, word in enumerate(labels):
            tok = [first + chars.index(c) if c in chars else (2 if ctc else 0) for c in word]          # [UNK] = 2 (CTC) / 0 (Attn)
            case = (bi, si)
            if case == (0, 1):                               # one wrong character
                tok = tok[:2] + [first + 30] + tok[3:]
            if ctc:
                if case == (0, 2):
                    tok = []                                 # all blanks: empty prediction
                rows.append(ctc_row(tok))
            else:
                if case == (0, 2):
                    rows.append(attn_row([]))                # [EOS] first: empty prediction
                elif case == (1, 1):
                    rows.append(([first + 1, first + 2] * T)[:T])      # no [EOS] at all
                else:
                    rows.append(attn_row(tok))
        tgt = torch.tensor(rows, dtype=torch.long)           # [B,T]
        lg = torch.from_numpy(W.uniform(f"crafted:{kind}:{bi}", (len(labels), T, C), -1.0, 1.0, 77))
        lg.scatter_add_(2, tgt.unsqueeze(2), torch.full((len(labels), T, 1), 4.0))
        image = torch.zeros(len(labels), 4, 32, 256)
        batches.append((image, labels))
        logits.append(lg)
    return chars, batches, logits


def fake_text_samples(path, seed=5):
    """deterministic (images, labels) for a dataset directory path such as "rootA/Latin": RGBA uint8 crops of varying size and
    labels over a small alphabet, a few of them longer than batch_max_length (the readers must filter those out)"""
    lang = path.rstrip("/").split("/")[-1]
    n = 37 + 11 * (sum(ord(c) for c in path) % 5)
    alphabet = "abcdefghijklmnopqrstuvwxyz" + lang.lower()
    images, labels = [], []
    for i in range(n):
        name = f"fake:{path}:{i}"
        h = int(W.randint(name + ":h", (1,), 20, 41, seed)[0])
        w = int(W.randint(name + ":w", (1,), 60, 201, seed)[0])
        images.append((W.uniform(name, (h, w, 4), 0.0, 255.999, seed)).astype(np.uint8))
        L = int(W.randint(name + ":len", (1,), 1, 31, seed)[0])          # up to 30 characters: some exceed 25
        ids = W.randint(name + ":lab", (L,), 0, len(alphabet), seed)
        labels.append("".join(alphabet[j] for j in ids))
    return images, labels

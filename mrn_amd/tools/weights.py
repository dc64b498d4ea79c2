"""Deterministic, torch-independent synthetic tensors: value = f(name, shape, seed).

Used to give the reference model (when golden vectors are generated), the CPU oracle and the HIP-backed modules
byte-identical weights and inputs without shipping weight blobs: numpy Philox keyed by a hash of the name.
"""
import hashlib

import numpy as np


def _rng(name, seed):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.Philox(key=int.from_bytes(h[:8], "little")))


def uniform(name, shape, lo=-1.0, hi=1.0, seed=0):
    u = _rng(name, seed).random(size=tuple(shape), dtype=np.float64)
    return (lo + (hi - lo) * u).astype(np.float32)


def randint(name, shape, lo, hi, seed=0):
    """integers in [lo, hi)"""
    u = _rng(name, seed).random(size=tuple(shape), dtype=np.float64)
    return np.minimum((lo + np.floor(u * (hi - lo))).astype(np.int64), hi - 1)


def smooth_image(name, shape, seed=0, terms=6):
    """[B,C,H,W] low-frequency field in [-1,1] (sum of a few cosines).  Parity inputs use this instead of white
    noise because the TPS grid P_hat.(inv_delta_C.C') is ill-conditioned in fp32 (~1e-5 absolute in the grid, i.e.
    ~2e-3 pixel): on white noise that alone moves the reference's own features by 6e-3 relative between two
    summation orders, which says nothing about an implementation.  Text crops are smooth at that scale."""
    B, C, H, W = shape
    r = _rng(name, seed)
    amp = r.random((B, C, terms)) + 0.2
    fx = r.random((B, C, terms)) * 4.0
    fy = r.random((B, C, terms)) * 1.5
    ph = r.random((B, C, terms)) * 2 * np.pi
    y = (np.arange(H) / H)[None, None, None, :, None]
    x = (np.arange(W) / W)[None, None, None, None, :]
    f = (amp[..., None, None] * np.cos(2 * np.pi * (fx[..., None, None] * x + fy[..., None, None] * y) + ph[..., None, None])).sum(2)
    f = f / amp.sum(2)[..., None, None]
    return f.astype(np.float32)


def fiducial_bias(num_fid):
    """The RARE initial fiducial layout (top row y: 0..-1, bottom row y: 1..0), flattened [F*2]."""
    half = num_fid // 2
    xs = np.linspace(-1.0, 1.0, half)
    top = np.stack([xs, np.linspace(0.0, -1.0, half)], 1)
    bot = np.stack([xs, np.linspace(1.0, 0.0, half)], 1)
    return np.concatenate([top, bot], 0).reshape(-1).astype(np.float32)


def det_param(key, shape, seed=0):
    """A plausible value for the state_dict entry `key` (keeps activations O(1) through deep ReLU/BN stacks)."""
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_var":
        return uniform(key, shape, 0.5, 1.5, seed)
    if leaf == "running_mean":
        return uniform(key, shape, -0.1, 0.1, seed)
    if key.endswith("localization_fc2.weight"):
        return uniform(key, shape, -0.02, 0.02, seed)
    if key.endswith("localization_fc2.bias"):
        return fiducial_bias(shape[0] // 2) + uniform(key, shape, -0.05, 0.05, seed)
    if "char_embeddings" in key:
        return uniform(key, shape, -1.0, 1.0, seed)
    if ".rnn." in key:                      # LSTM / LSTMCell: torch default U(-1/sqrt(H), 1/sqrt(H))
        hidden = shape[0] // 4
        b = 1.0 / np.sqrt(hidden)
        return uniform(key, shape, -b, b, seed)
    if len(shape) <= 1:
        if leaf == "weight":                # norm scales
            return uniform(key, shape, 0.5, 1.5, seed)
        return uniform(key, shape, -0.1, 0.1, seed)
    fan_in = int(np.prod(shape[1:]))
    gain = np.sqrt(2.0) if len(shape) == 4 else 1.0
    b = gain * np.sqrt(3.0 / fan_in)
    return uniform(key, shape, -b, b, seed)


def fill_state_dict(sd, seed=0):
    """In-place: overwrite every tensor of a torch state_dict with det_param(key, shape). Aliased entries (same
    storage under two keys, e.g. `fc` and `Prediction.generator`) get the value of the key visited last -- so visit
    in sorted order for reproducibility."""
    import torch

    for key in sorted(sd.keys()):
        t = sd[key]
        v = det_param(canonical_key(key), t.shape, seed)
        t.copy_(torch.from_numpy(v).to(t.dtype))
    return sd


def canonical_key(key):
    """Aliases share one value: `<pre>Prediction.generator.*` (Attn) and `<pre>Prediction.*` (CTC head) are `<pre>fc.*`."""
    key = key.replace("Prediction.generator.", "fc.")
    if key.endswith("Prediction.weight") or key.endswith("Prediction.bias"):
        key = key.replace("Prediction.", "fc.")
    if key.startswith("module."):
        key = key[len("module."):]
    return key

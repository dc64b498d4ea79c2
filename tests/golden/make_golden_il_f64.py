"""The float64 yardstick of the TRBA / CRNN learner flows: the REFERENCE learners of make_golden_il.py / make_golden_il2.py (LwF, EWC, DER,
WA, Joint driven through incremental_train() over two tasks) run once more in float64 arithmetic -- same deterministic weights
(fp32 values, widened), same deterministic batches -- and only the parameter movement of the task's optimiser steps is stored:

    python tests/golden/make_golden_il_f64.py            # -> tests/golden/il_{trba,crnn}_f64.npz   (build container only)

Why: Adam normalises every element's update to ~lr, so after two steps the movement of elements whose gradient is near zero has an
essentially arbitrary sign in ANY fp32 implementation.  tests/test_il_golden_gpu.py therefore judges the HIP movement against this
float64 run and requires it to be as close to it as (a small multiple of) the reference's own fp32 run is -- instead of a fixed
50 % relative-L2 band against the fp32 reference.

Harness shims on top of make_golden_il.py's (all confined to this script): torch's default dtype is float64, torch.FloatTensor
(the attention decoder's hidden-state buffers, modules/prediction.py) is torch.DoubleTensor, the deterministic loader hands
out float64 crops, and the model is widened with .double() before the deterministic fill (the reference builds its TPS constants
with .float(): same fp32 VALUES, float64 arithmetic on them).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import make_golden_il as G  # noqa: E402   (imports the reference with its stubs)
from tests.golden import make_golden_il2 as G2  # noqa: E402
from tests import helpers  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class DetLoader64(helpers.DetLoader):
    def _batch(self, n):
        image, labels = super()._batch(n)
        return image.double(), labels


def main():
    import tempfile
    torch.set_default_dtype(torch.float64)
    torch.FloatTensor = torch.DoubleTensor
    G.DetLoader = DetLoader64
    G2.DetLoader = DetLoader64
    fill = G.det_fill

    def det_fill64(learner, seed):
        learner.model.double()        # (the reference casts the TPS constants and one bias with .float(): widen them back, values unchanged)
        for m in learner.model.modules():      # (on a GPU-less host the grid constants are plain attributes, transformation.py:137-146)
            for name in ("inv_delta_C", "P_hat"):
                if isinstance(getattr(m, name, None), torch.Tensor):
                    setattr(m, name, getattr(m, name).double())
        fill(learner, seed)
    G.det_fill = det_fill64
    for kind in ("trba", "crnn"):
        d = {}
        tmp = tempfile.mkdtemp()
        cwd = os.getcwd()
        os.chdir(tmp)
        os.makedirs("./saved_models/g", exist_ok=True)
        try:
            for which in ("lwf", "ewc", "der"):
                for k, v in G.run_learner(kind, which).items():
                    if "/delta/" in k or k.endswith("param_keys") or k.endswith("/losses"):
                        d[f"{which}/{k}"] = v
                for f in os.listdir("./saved_models/g"):
                    os.remove(os.path.join("./saved_models/g", f))
            for which, fn in (("wa", G2.run_wa), ("joint", G2.run_joint)):
                for k, v in fn(kind).items():
                    if "/delta/" in k or k.endswith("param_keys") or k.endswith("/losses"):
                        d[f"{which}/{k}"] = v
                for f in os.listdir("./saved_models/g"):
                    os.remove(os.path.join("./saved_models/g", f))
        finally:
            os.chdir(cwd)
        np.savez_compressed(os.path.join(OUT, f"il_{kind}_f64.npz"), **d)
        print(f"il_{kind}_f64.npz:", len(d), "arrays")


if __name__ == "__main__":
    main()

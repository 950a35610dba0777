"""Generates tests/golden/adan.npz from the reference's OWN optimizer (/root/reference/optimizer.py:23-249), run in the
build container only:   python tests/golden/make_golden_adan.py
Two parameter groups with the reference's hyper-parameters (main.py:132: lr 5e-3 / 5e-2, eps 1e-8, weight_decay 2e-5,
max_grad_norm 5.0, foreach=False), six steps of seeded gradients, one of them large enough to trigger the global clip."""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_optimizer", "/root/reference/optimizer.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def main():
    g = torch.Generator().manual_seed(3)
    table = torch.randn(257, generator=g)
    w = torch.randn(5, 7, generator=g)
    grads = [(torch.randn(257, generator=g) * s, torch.randn(5, 7, generator=g) * s) for s in (1e-3, 0.2, 9.0, 0.05, 1e-4, 2.0)]
    p1, p2 = torch.nn.Parameter(table.clone()), torch.nn.Parameter(w.clone())
    opt = ref.Adan([{"params": [p1], "lr": 5e-2}, {"params": [p2], "lr": 5e-3}], eps=1e-8, weight_decay=2e-5,
                   max_grad_norm=5.0, foreach=False)
    out = {"table0": table.numpy(), "w0": w.numpy()}
    for i, (g1, g2) in enumerate(grads):
        p1.grad, p2.grad = g1.clone(), g2.clone()
        opt.step()
        out[f"g1_{i}"], out[f"g2_{i}"] = g1.numpy(), g2.numpy()
        out[f"table_{i}"], out[f"w_{i}"] = p1.detach().numpy().copy(), p2.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "adan.npz"), **out)
    print("adan.npz written:", {k: v.shape for k, v in out.items() if k.endswith("_5")})


if __name__ == "__main__":
    main()

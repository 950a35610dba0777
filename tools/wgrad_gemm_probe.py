"""Where the reference-shaped step's time goes (profiles/kernel_stats_r06_reference_shaped.csv): 1.39 of its 1.79 s are
the weight-gradient GEMMs of torch's nn.Linear backward - dW = dY^T X with K = the ~11 M sample rows and a 4 x 64 / 64 x 64
/ 64 x 32 output: hipBLASLt picks a kernel without split-K (one 16 x 16 tile walks all 11 M rows: 82 ms per call).  This
probe times those three shapes (binary16, as under autocast) under torch's two BLAS back ends, and a plain fp32
reduction formulation, so INTEGRATION.md can say what a user of the UNCHANGED reference route can do about it.

    python tools/wgrad_gemm_probe.py --out gpurun_out/wgrad_gemm_probe.json"""
import argparse
import json
import os

import torch


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_878_464)
    ap.add_argument("--out", default="gpurun_out/wgrad_gemm_probe.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.rows
    res = {"rows": n, "env_TORCH_BLAS_PREFER_HIPBLASLT": os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"), "ms": {}}
    shapes = {"dW3 = dY[n,4]^T H2[n,64]": (4, 64), "dW2 = dH2[n,64]^T H1[n,64]": (64, 64), "dW1 = dH1[n,64]^T X[n,32]": (64, 32)}
    for backend in ("default", "cublas", "cublaslt"):
        try:
            if backend != "default":
                torch.backends.cuda.preferred_blas_library(backend)
            res["ms"][backend] = {"library": str(torch.backends.cuda.preferred_blas_library())}
            for name, (o, i) in shapes.items():
                dy = torch.randn(n, o, device=dev, dtype=torch.float16)
                x = torch.randn(n, i, device=dev, dtype=torch.float16)
                res["ms"][backend][name] = timeit(lambda: dy.t() @ x)
                # the dgrad / forward shapes beside it, for scale
                w = torch.randn(o, i, device=dev, dtype=torch.float16)
                res["ms"][backend][name + " | forward X W^T"] = timeit(lambda: x @ w.t())
                del dy, x, w
        except Exception as e:  # noqa: BLE001
            res["ms"][backend] = {"error": repr(e)}
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

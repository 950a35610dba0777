"""Adan (Adaptive Nesterov Momentum) as the reference configures it (main.py:132: lr 5e-3 / 5e-2 for the grid,
eps 1e-8, weight_decay 2e-5, max_grad_norm 5.0, betas (0.98, 0.92, 0.99), no_prox False) - the update rule of
/root/reference/optimizer.py:100-255 restated with two differences that do not change the math:
the global-norm clip factor stays on the device (the reference's `.item()` costs a host sync per step), and on the GPU
the whole update of a parameter tensor is ONE fused kernel (csrc/optim.hip, C ABI Part 6: mi3d_sumsq_accumulate +
mi3d_adan_step - SURVEY 8(f3)) instead of ~12 elementwise passes over the 48.8 MB table.  CPU tensors (the golden
trajectory test against the reference's own optimizer.py) take the same arithmetic through torch ops."""
import math

import torch


class Adan(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0,
                 no_prox=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      max_grad_norm=max_grad_norm, no_prox=no_prox))

    def _fused_ok(self):
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        return bool(ps) and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                                and p.grad.dtype == torch.float32 and p.data_ptr() % 16 == 0
                                and p.grad.data_ptr() % 16 == 0 for p in ps)

    def _fused_step(self):
        import ctypes as C
        from . import _lib as L
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        dev = ps[0].device
        max_norm = float(self.defaults["max_grad_norm"])
        sumsq = None
        with torch.cuda.device(dev):
            if max_norm > 0:
                sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
                for p in ps:
                    L.call("mi3d_sumsq_accumulate", L.ptr(p.grad), C.c_size_t(p.grad.numel()), L.ptr(sumsq), L.stream(p))
            for group in self.param_groups:
                b1, b2, b3 = group["betas"]
                group["step"] = group.get("step", 0) + 1
                k = group["step"]
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    first = len(st) == 0 or "neg_pre_grad" not in st or k == 1
                    if len(st) == 0:
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                        st["exp_avg_diff"] = torch.zeros_like(p)
                    if "neg_pre_grad" not in st:
                        st["neg_pre_grad"] = torch.empty_like(p)
                    L.call("mi3d_adan_step", L.ptr(p), L.ptr(p.grad), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]),
                           L.ptr(st["exp_avg_diff"]), L.ptr(st["neg_pre_grad"]), C.c_size_t(p.numel()), L.ptr(sumsq),
                           max_norm, float(self.param_groups[-1]["eps"]), int(first), b1, b2, b3, 1 - b1 ** k,
                           1 - b2 ** k, math.sqrt(1 - b3 ** k), group["lr"], group["weight_decay"], group["eps"],
                           int(bool(group["no_prox"])), L.stream(p))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._fused_ok():
            self._fused_step()
            return loss
        clip = None
        if self.defaults["max_grad_norm"] > 0:
            grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
            if grads:
                total = torch.sqrt(sum(g.pow(2).sum() for g in grads))
                clip = torch.clamp(self.defaults["max_grad_norm"] / (total + self.param_groups[-1]["eps"]), max=1.0)
        for group in self.param_groups:
            b1, b2, b3 = group["betas"]
            group["step"] = group.get("step", 0) + 1
            k = group["step"]
            bc1, bc2, bc3s = 1 - b1 ** k, 1 - b2 ** k, math.sqrt(1 - b3 ** k)
            lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if clip is not None:
                    g.mul_(clip)
                st = self.state[p]
                if len(st) == 0:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                    st["exp_avg_diff"] = torch.zeros_like(p)
                if "neg_pre_grad" not in st or k == 1:
                    st["neg_pre_grad"] = g.clone().neg_()  # already clipped: -(clip * grad)
                m, n, d, npg = st["exp_avg"], st["exp_avg_sq"], st["exp_avg_diff"], st["neg_pre_grad"]
                npg.add_(g)                                    # g_t - g_{t-1}
                m.mul_(b1).add_(g, alpha=1 - b1)
                d.mul_(b2).add_(npg, alpha=1 - b2)
                npg.mul_(b2).add_(g)                           # g_t + b2 (g_t - g_{t-1})
                n.mul_(b3).addcmul_(npg, npg, value=1 - b3)
                denom = (n.sqrt() / bc3s).add_(eps)
                if group["no_prox"]:
                    p.mul_(1 - lr * wd)
                    p.addcdiv_(m, denom, value=-lr / bc1)
                    p.addcdiv_(d, denom, value=-lr * b2 / bc2)
                else:
                    p.addcdiv_(m, denom, value=-lr / bc1)
                    p.addcdiv_(d, denom, value=-lr * b2 / bc2)
                    p.div_(1 + lr * wd)
                npg.zero_().add_(g, alpha=-1.0)
        return loss

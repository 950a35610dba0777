"""Data parallelism for the SDS loop: one process per GPU, one novel view per rank per step, and ONE flat
all-reduce of the NeRF parameter gradients per step.

The reference has no working multi-GPU path (vestigial DDP wrap, nerf/utils.py:255-264); this is new.
All parameter gradients live as views into a single contiguous bucket (12 196 240 + 6 532 fp32 = 48.8 MB for the
default field), so the collective is a single in-place RCCL all-reduce over xGMI - no per-tensor launches, no copy
in or out - issued after backward and before the optimizer's global-norm clip, which therefore sees identical
gradients on every rank.  Works with any torch.distributed backend (tests run it on gloo, world_size 2).
"""
import torch
import torch.distributed as dist

# how many collectives this process issued (bench.py reports them; tests/test_rccl_gpu.py checks the RCCL path really ran)
STATS = {"all_reduce": 0, "broadcast": 0}


def _active(group=None):
    """Collectives run whenever a process group exists - also at world size 1 (one rank's all-reduce is a copy onto
    itself), so that a single-GPU launch under torch.distributed.run exercises exactly the calls an 8-GPU run makes."""
    return dist.is_available() and dist.is_initialized()


class FlatGradBucket:
    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all bucketed parameters must share device and dtype")
            p.grad = self.flat[off:off + p.numel()].view_as(p)  # autograd accumulates in place into the bucket
            off += p.numel()
        # this bucket reads `.grad` from Python (GridParameter.grad completes a parked scatter first): the deferred point-0
        # scatter stays on under a process group (grid_ops.deferral_allowed)
        from . import grid_ops
        grid_ops.PYTHON_GRAD_SYNC = True

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def _check_views(self):
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                # something replaced .grad (e.g. zero_grad(set_to_none=True)): fold it back into the bucket
                view = self.flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                else:
                    view.zero_()
                p.grad = view
            off += p.numel()

    def zero(self):
        self._check_views()
        self.flat.zero_()

    def all_reduce_mean(self):
        """Average the bucket over the ranks (no-op without an initialised process group)."""
        self._check_views()
        if not _active(self.group):
            return
        world = dist.get_world_size(self.group)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        STATS["all_reduce"] += 1
        if world > 1:
            self.flat.div_(world)


def broadcast_module_state(module, src=0, process_group=None):
    """Same initial weights, buffers (density grid / bitfield) on every rank."""
    if not _active(process_group):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)
            STATS["broadcast"] += 1


def sync_occupancy(model, src=0, process_group=None):
    """After `update_extra_state` on rank `src` (it draws torch.rand jitter): ship the 262 144-byte bitfield
    and the mean density to the other ranks every 16 steps."""
    if not _active(process_group):
        return
    dist.broadcast(model.density_bitfield, src=src, group=process_group)
    dist.broadcast(model.density_grid, src=src, group=process_group)
    md = torch.tensor([float(model.mean_density)], device=model.density_bitfield.device)
    dist.broadcast(md, src=src, group=process_group)
    STATS["broadcast"] += 3
    model.mean_density = float(md.item())

"""torch.optim.Adam's update (what the reference steps both networks with, codes/models/SRRaGAN_model.py:147-160) as ONE kernel launch over
all tensors of a parameter group (esr_adam_*, csrc/esr_optim.hip).  Same state layout as torch.optim.Adam ('step', 'exp_avg', 'exp_avg_sq'
per parameter), so schedulers, state_dict() and checkpoints written by either implementation load into the other."""
import ctypes as C
import math

import torch

from . import _lib
from .act import require_gpu, stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}           # group index -> {pointer fingerprint: (fingerprint, workspace tensor, n tensors, n chunks, pinned staging block)}
        self.table_uploads = 0      # (diagnostic: how often a table had to be built and copied)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            grads = []
            for p in ps:
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                grads.append(g)
                st = self.state[p]
                if not st:
                    require_gpu(p, 'parameter')
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise _lib.EsrError('esr_hip.optim.Adam steps contiguous fp32 parameters')
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # every tensor that has a gradient steps together (a parameter that sat out earlier steps would need its own bias
            # corrections): the group's tensors SHARE one 'step' tensor object, re-established after load_state_dict's deep copies
            first = self.state[ps[0]]['step']
            if any(self.state[p]['step'] is not first for p in ps):        # (identity checks of ~700 objects: microseconds)
                if any(float(self.state[p]['step']) != float(first) for p in ps):
                    raise _lib.EsrError('parameters of one group have stepped a different number of times: put them into separate groups')
                for p in ps:
                    self.state[p]['step'] = first
            t = float(first) + 1
            # everything the uploaded table points at: parameters, gradients AND both moment tensors (load_state_dict replaces the latter)
            fp = tuple((p.data_ptr(), g.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr(), p.numel()) for p, g in zip(ps, grads))
            # a few tables per group, keyed by the pointer fingerprint: gradient storages typically alternate between two or three addresses
            # (the allocator hands back the blocks of the step before last), and a table that is still on the device costs nothing
            cache = self._tables.setdefault(gi, {})
            tab = cache.get(fp)
            if tab is None:
                arr = (_lib.AdamTensor * len(ps))()
                for a, p, g in zip(arr, ps, grads):
                    st = self.state[p]
                    a.p, a.g, a.m, a.v, a.n = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
                need = _lib.lib.esr_adam_workspace_bytes(arr, len(ps))
                _lib.check(min(need, 0), 'esr_adam_workspace_bytes')
                ws = None
                if len(cache) >= 4:
                    old = cache.pop(next(iter(cache)))    # oldest entry: its device workspace is taken over when it is large enough
                    if old[1].numel() >= int(need) and old[1].device == ps[0].device:
                        ws = old[1]                       # (stream-ordered: the table copy below queues behind the launch that last read it; the pinned
                                                          # block is NOT reused — the host would overwrite it while an earlier copy may still be queued)
                if ws is None:
                    ws = torch.empty(int(need), dtype=torch.uint8, device=ps[0].device)
                # the table goes through PINNED host memory and an asynchronous copy on the step's stream: gradient tensors are new storages
                # every step more often than not (autograd hands over the buffers the backward pass produced), and a blocking upload here
                # stalled the host until the whole backward pass had run — the next step's launches then started from an idle GPU
                # (configs[2] step: ~1.5 ms of gaps behind the two Adam launches).  The pinned block stays referenced by the table entry;
                # torch's pinned-memory allocator does not recycle it before the copy has executed.
                pin = torch.empty(int(need), dtype=torch.uint8, pin_memory=True)
                nchunks = _lib.lib.esr_adam_table(arr, len(ps), pin.data_ptr(), pin.numel())
                _lib.check(min(nchunks, 0), 'esr_adam_table')
                ws[:int(need)].copy_(pin, non_blocking=True)
                tab = cache[fp] = (fp, ws, len(ps), int(nchunks), pin)
                self.table_uploads += 1
            b1, b2 = group['betas']
            _lib.check(_lib.lib.esr_adam_run(tab[1].data_ptr(), tab[2], tab[3], float(group['lr']), b1, b2, group['eps'], group['weight_decay'],
                                             1 - b1 ** t, math.sqrt(1 - b2 ** t), stream_ptr()), 'esr_adam_run')
            first.fill_(t)
            # the kernel wrote the parameters behind torch's back: tell the version counters (weight packs, autograd's saved-tensor checks)
            torch.autograd.graph.increment_version(ps)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}           # the loaded moment tensors are new storages

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, '_tables'):
            self._tables = {}

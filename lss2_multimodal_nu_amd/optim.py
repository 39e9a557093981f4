"""ClipAdam: `clip_grad_norm_` + `torch.optim.Adam.step()` of the reference's loop (train.py:41, 62-63) as three HIP
launches (csrc/optim.hip) instead of the 17 the ATen ops take.

    opt = ClipAdam(model.parameters(), lr=1e-3, weight_decay=1e-7, max_grad_norm=5.0)
    loss.backward(); opt.step()            # clips to max_grad_norm, then Adam; opt.grad_norm = norm before the clip

Same update rule and defaults as torch.optim.Adam (amsgrad / maximize / per-group foreach switches are not offered:
the reference does not use them), same `state_dict()` layout per parameter ('step', 'exp_avg', 'exp_avg_sq'), so a
checkpoint written with one loads into the other.  The step counter, the norm and the clip coefficient live on the
device: `step()` never synchronises and can be captured in a HIP graph (dp.GraphedTrainStep).  There is no CPU path:
parameters and gradients must be fp32 CUDA tensors (contiguous), and the HIP library must be present."""
import ctypes

import torch

from . import _native as N


class _OptTensor(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_longlong)]


class ClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("ClipAdam: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self._dev = {}       # device -> (state floats [8], partials buffer)
        self.grad_norm = None  # 0-dim device tensor: total gradient norm BEFORE the clip of the latest step

    def _device_state(self, device, n_partials):
        st = self._dev.get(device)
        if st is None or st[1].numel() < n_partials:
            state = st[0] if st is not None else torch.zeros(8, dtype=torch.float32, device=device)
            st = self._dev[device] = (state, torch.empty(max(n_partials, 1), dtype=torch.float32, device=device))
        return st

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """Clip the total gradient norm of ALL parameter groups to `max_grad_norm` (default: the constructor's; None or
        <= 0: no clip), then one Adam step.  The norm is taken over every parameter that has a gradient, as
        clip_grad_norm_(model.parameters(), ...) does; groups may differ in lr / betas / eps / weight_decay."""
        if closure is not None:
            raise ValueError("ClipAdam.step takes no closure (the loss is computed by the caller)")
        clip = self.max_grad_norm if max_grad_norm is None else max_grad_norm
        clip = float(clip) if clip is not None and clip > 0 else 0.0
        groups = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if ps:
                groups.append((group, ps))
        if not groups:
            return None
        device = groups[0][1][0].device
        entries, sizes = [], []
        for group, ps in groups:
            for p in ps:
                g = p.grad
                if (not p.is_cuda or p.device != device or p.dtype != torch.float32 or g.dtype != torch.float32
                        or not p.is_contiguous() or not g.is_contiguous() or g.is_sparse):
                    raise ValueError("ClipAdam: parameters and gradients must be contiguous fp32 tensors on one GPU")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                entries.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
                sizes.append(p.numel())
        n_partials = int(N.lib().lss_clip_adam_partials((ctypes.c_longlong * len(sizes))(*sizes), len(sizes)))
        fresh_counter = device not in self._dev
        state, partials = self._device_state(device, n_partials)
        if fresh_counter:
            # a loaded checkpoint (this class's or torch.optim.Adam's) carries its step count per parameter: the shared
            # device counter continues from it (read once, at the first step after the load: never inside a capture)
            loaded = [self.state[p]["step"] for p, _, _, _ in entries if "step" in self.state[p]]
            if loaded:
                state[0] = float(max(float(t) for t in loaded))
        for p, _, _, _ in entries:           # 'step' of every parameter is the one shared device counter
            self.state[p]["step"] = state[0]
        # one launch group per distinct hyper-parameter set; the norm of the FIRST call covers every tensor
        hyper0 = None
        same = all((g["lr"], g["betas"], g["eps"], g["weight_decay"]) ==
                   (groups[0][0]["lr"], groups[0][0]["betas"], groups[0][0]["eps"], groups[0][0]["weight_decay"])
                   for g, _ in groups)
        if not same:
            raise ValueError("ClipAdam: parameter groups with different hyper-parameters are not supported (the "
                             "reference builds one group: train.py:41)")
        hyper0 = groups[0][0]
        tab = (_OptTensor * len(entries))()
        for i, (p, g, m, v) in enumerate(entries):
            tab[i].p, tab[i].g, tab[i].m, tab[i].v, tab[i].n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        N.check(N.lib().lss_clip_adam_step(ctypes.cast(tab, ctypes.c_void_p), len(entries), state.data_ptr(),
                                           partials.data_ptr(), n_partials, float(hyper0["lr"]),
                                           float(hyper0["betas"][0]), float(hyper0["betas"][1]), float(hyper0["eps"]),
                                           float(hyper0["weight_decay"]), clip, N.stream()), "lss_clip_adam_step")
        self.grad_norm = state[1]
        return None

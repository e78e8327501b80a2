"""Adam for the mapper's and tracker's loops on the HIP path.

Drop-in for the way the reference uses ``torch.optim.Adam``:

    optimizer = Adam([{'params': decoders_para_list, 'lr': 0}, {'params': planes_para, 'lr': 0}, ...])   # Mapper.py:291-299
    optimizer.param_groups[1]['lr'] = cfg['mapping']['lr']['planes_lr'] * lr_factor                      # Mapper.py:301-306
    optimizer.zero_grad(); loss.backward(); optimizer.step()                                             # Mapper.py:348-350

(same for ``src/Tracker.py:262-266,206-208``).  Same constructor keywords, ``param_groups``, per-parameter ``state``
with torch's key names (``step``, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` is interchangeable with
``torch.optim.Adam``'s.  One ``eslam_adam_step`` launch updates every tensor of the step (csrc/eslam_adam.hip);
there is no CPU implementation: parameters must live on the GPU.

Extras over the reference's usage (both off by default):
  fused_zero_grad=True  the step clears the gradients it consumed and ``zero_grad()`` keeps the (now zero) ``.grad``
                        tensors, so no separate 27-70 MB fill is needed;
  capturable=True       the step count lives in a device counter, so ``step()`` can be captured into a hipGraph and
                        replayed (``state['step']`` then mirrors the number of host-side ``step()`` calls only).
"""
import ctypes

import torch

from . import _hip


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *,
                 maximize=False, fused_zero_grad=False, capturable=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise ValueError("myslam_amd.optim.Adam implements the reference's configuration only: "
                             "weight_decay=0, amsgrad=False, maximize=False")
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid Adam hyper-parameters: lr={lr}, betas={betas}, eps={eps}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False)
        super().__init__(params, defaults)
        self.fused_zero_grad = bool(fused_zero_grad)
        self.capturable = bool(capturable)
        self._step_dev = None
        self._table_cache = {}
        self._fast = None            # (params, pointer signature, hyper signature, table, step tensors, step, device)

    # -- helpers ------------------------------------------------------------------------------------------------
    @staticmethod
    def _dense_like(p, g):
        """True when g enumerates its elements in the same memory order as p (then the update is elementwise over
        the two storages)."""
        return g.shape == p.shape and g.stride() == p.stride()

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            # CPU scalar, as torch.optim.Adam keeps it; parameters initialised in the same step() share one tensor
            # (one host-side increment per step instead of 26)
            st["step"] = self._new_step
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._fast = None
        self._table_cache.clear()
        if self._step_dev is not None:
            raise RuntimeError("load_state_dict on a capturable optimiser that has already stepped is not supported")

    def zero_grad(self, set_to_none=True):
        if self.fused_zero_grad:
            return                      # step() already cleared what it consumed; .grad tensors stay allocated
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _hip.lib()
        # steady state of the mapper's inner loop: same tensors, same gradient buffers as the previous step ->
        # reuse the descriptor table, skip the per-parameter checks
        fast = self._fast
        if fast is not None:
            params, ptr_sig, hyper_sig, arr, step_tensors, step, dev = fast
            if ptr_sig == [x for p in params for x in (p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0)] \
                    and hyper_sig == [(g["lr"], g["betas"], g["eps"], len(g["params"])) for g in self.param_groups]:
                for t in step_tensors:
                    t += 1
                step += 1
                self._fast = (params, ptr_sig, hyper_sig, arr, step_tensors, step, dev)
                with _hip.on_device(dev):
                    _hip.check(lib.eslam_adam_step(ctypes.cast(arr, ctypes.c_void_p), len(arr), step,
                                                   _hip.ptr(self._step_dev) if self.capturable else None,
                                                   hyper_sig[0][1][0], hyper_sig[0][1][1], hyper_sig[0][2],
                                                   1 if self.fused_zero_grad else 0, _hip.stream_handle(dev)),
                               "eslam_adam_step")
                return loss
            self._fast = None
        # tensors are batched per (betas, eps, step): one launch in the reference's usage
        batches = {}
        self._new_step = torch.tensor(0.0, dtype=torch.float32)
        bumped = set()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                _hip.require_gpu_f32("Adam parameter", p)
                _hip.require_gpu_f32("Adam gradient", g)
                if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                    raise RuntimeError("Adam parameters must be dense (contiguous or channels_last)")
                if not self._dense_like(p, g):
                    # a gradient in another layout (e.g. NCHW for a channels_last plane): re-lay it once, like the param
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    p.grad = g
                st = self._init_state(p)
                if id(st["step"]) not in bumped:
                    st["step"] += 1
                    bumped.add(id(st["step"]))
                key = (float(b1), float(b2), float(group["eps"]), int(st["step"]), p.device.index)
                batches.setdefault(key, []).append((p, g, st, float(group["lr"])))
        for (b1, b2, eps, step, dev_index), items in batches.items():
            dev = torch.device("cuda", dev_index)
            sig = tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), lr)
                        for p, g, st, lr in items)
            arr = self._table_cache.get(sig)
            if arr is None:
                if len(self._table_cache) > 64:
                    self._table_cache.clear()
                arr = (_hip.AdamTensor * len(items))()
                for k, (pp, gp, mp, vp, n, lr) in enumerate(sig):
                    arr[k].param, arr[k].grad, arr[k].exp_avg, arr[k].exp_avg_sq, arr[k].n, arr[k].lr = pp, gp, mp, vp, n, lr
                self._table_cache[sig] = arr
            step_dev = None
            if self.capturable:
                if self._step_dev is None:
                    self._step_dev = torch.full((1,), step - 1, dtype=torch.int32, device=dev)
                step_dev = _hip.ptr(self._step_dev)
            with _hip.on_device(dev):
                _hip.check(lib.eslam_adam_step(ctypes.cast(arr, ctypes.c_void_p), len(items), step, step_dev, b1, b2, eps,
                                               1 if self.fused_zero_grad else 0, _hip.stream_handle(dev)),
                           "eslam_adam_step")
        if len(batches) == 1:
            (key, items), = batches.items()
            every = [p for g in self.param_groups for p in g["params"]]
            if len(items) == len(every):         # all parameters have gradients and share betas / eps / step
                uniq = list({id(st["step"]): st["step"] for _, _, st, _ in items}.values())
                self._fast = (every, [x for p in every for x in (p.data_ptr(), p.grad.data_ptr())],
                              [(g["lr"], g["betas"], g["eps"], len(g["params"])) for g in self.param_groups],
                              self._table_cache[tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                       st["exp_avg_sq"].data_ptr(), p.numel(), lr)
                                                      for p, g, st, lr in items)],
                              uniq, key[3], torch.device("cuda", key[4]))
        return loss

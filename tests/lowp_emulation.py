"""TEST INFRASTRUCTURE ONLY — the precision floor of a 16-bit compute mode, emulated on the pinned CPU oracle.

``LowpPolicy`` is a torch-function mode under which the oracle's matmul-class operands (linear / bmm / matmul /
conv2d inputs) are rounded to a 16-bit type (what an MFMA consumes) while every accumulation, statistic and
elementwise op stays fp32 ("operands only": the best any implementation with 16-bit MFMA operands can do), and,
optionally, the outputs of ops inside named regions are rounded as well (16-bit activation storage).  The routers,
the Block fusion and the losses are exempt, as on the HIP path (fp32 end to end there).

Tests print / use this floor next to the error of the HIP path so that a tolerance is never a guess:
    with LowpPolicy(torch.bfloat16): loss, logits, _ = O.forward(...)
"""
import threading

import torch
from torch.overrides import TorchFunctionMode

from oracle import d2r_oracle as O

_REGION = ["other"]
_lock = threading.Lock()
_installed = False


def _install():
    """Wraps the oracle's region-defining functions once so that the mode knows where it is."""
    global _installed
    with _lock:
        if _installed:
            return
        _installed = True

        def wrap(name, region):
            fn = getattr(O, name)

            def inner(*a, **k):
                _REGION.append(region)
                try:
                    return fn(*a, **k)
                finally:
                    _REGION.pop()
            inner.__wrapped__ = fn
            setattr(O, name, inner)

        for n, r in (("bert_layer", "enc"), ("clip_layer", "enc"), ("vision_embed", "enc"), ("text_embed", "enc"),
                     ("interaction_module", "routing"), ("block_fusion", "block"), ("router_gate", "router"),
                     ("js_div", "loss"), ("_saf", "saf")):
            wrap(n, r)
        for cname in list(O.CELLS):  # the routing layer looks its cells up through this table
            O.CELLS[cname] = getattr(O, "cell_" + cname)


_MM = {torch.nn.functional.linear, torch.bmm, torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__,
       torch.nn.functional.conv2d, torch.Tensor.bmm}
_EW = {torch.nn.functional.layer_norm, torch.nn.functional.relu, torch.tanh, torch.nn.functional.gelu, torch.sigmoid,
       torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.Tensor.mul, torch.Tensor.__mul__,
       torch.Tensor.__rmul__, torch.Tensor.sub, torch.Tensor.__sub__, torch.Tensor.pow, torch.Tensor.__truediv__,
       torch.Tensor.div, torch.cat, torch.nn.functional.embedding, torch.sqrt, torch.Tensor.sqrt}


class LowpPolicy(TorchFunctionMode):
    def __init__(self, lowp=torch.bfloat16, store=()):
        """store: regions ('enc', 'routing', 'saf', 'other') whose op OUTPUTS are rounded too (16-bit storage)."""
        super().__init__()
        _install()
        self.lowp, self.store = lowp, set(store)

    def _q(self, x):
        return x.to(self.lowp).to(x.dtype) if torch.is_tensor(x) and x.is_floating_point() else x

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        region = _REGION[-1]
        if region in ("router", "block", "loss"):
            return func(*args, **kwargs)
        if func in _MM:
            n = 3 if func is torch.nn.functional.conv2d else 2
            args = tuple(self._q(a) if i < n else a for i, a in enumerate(args))
        out = func(*args, **kwargs)
        if region in self.store and (func in _MM or func in _EW) and torch.is_tensor(out) and out.is_floating_point():
            if func in _MM and func is not torch.nn.functional.linear and out.shape[-1] < 40:
                return out  # attention scores stay fp32 inside the fused cores (small fixtures: Lk < 40)
            if out.dim() >= 2 and out.shape[-1] == 1:
                return out
            return self._q(out)
        return out


STORE_ALL = ("enc", "routing", "saf", "other")


def lowp_floor(sd, cfg, batch, train, lowp=torch.bfloat16, store=()):
    """(loss, logits) of the oracle in fp32 under the policy; ``sd`` fp32 CPU state dict, ``batch`` the 5-tuple."""
    ids, mask, tt, labels, images = batch
    sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad(), LowpPolicy(lowp, store):
        loss, logits, _ = O.forward(sd32, cfg, ids, mask, tt, labels, images.float(), train=train)
    return loss, logits

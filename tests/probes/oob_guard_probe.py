"""Out-of-bounds detector for every buffer the host side allocates with torch.empty / empty_like / zeros during one
forward + backward of the small two-branch model: each allocation gets 4 KiB of guard bytes on both sides, checked after the step.
A violated guard names the allocation site (file:line of the caller inside d2r_amd)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore

dev = torch.device("cuda:0")
G = 4096
live = []
_empty, _empty_like, _zeros = torch.empty, torch.empty_like, torch.zeros


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "d2r_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def guarded(shape, dtype, device, fill=None):
    if device is None or torch.device(device).type != "cuda":
        return None
    n = 1
    for s_ in shape:
        n *= int(s_)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    pad = (-nbytes) % 256
    buf = _empty(G + nbytes + pad + G, dtype=torch.uint8, device=device)
    buf[:G] = 0xA5
    buf[G + nbytes:] = 0xA5
    t = buf[G:G + nbytes].view(dtype).view(tuple(int(s_) for s_ in shape))
    if fill is not None:
        t.fill_(fill)
    live.append((buf, nbytes, _site(), tuple(shape), dtype))
    return t


def p_empty(*size, dtype=None, device=None, **kw):
    shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
    t = guarded(shape, dtype or torch.float32, device) if not kw.get("pin_memory") else None
    return t if t is not None else _empty(*size, dtype=dtype, device=device, **kw)


def p_empty_like(x, dtype=None, device=None, **kw):
    t = guarded(x.shape, dtype or x.dtype, device or x.device) if (device or x.device).type == "cuda" else None
    return t if t is not None else _empty_like(x, dtype=dtype, device=device, **kw)


def p_zeros(*size, dtype=None, device=None, **kw):
    shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
    t = guarded(shape, dtype or torch.float32, device, fill=0)
    return t if t is not None else _zeros(*size, dtype=dtype, device=device, **kw)


def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for buf, nbytes, site, shape, dtype in live:
        lo, hi = buf[:G], buf[G + nbytes:]
        if not bool((lo == 0xA5).all()) or not bool((hi == 0xA5).all()):
            bad += 1
            nlo, nhi = int((lo != 0xA5).sum()), int((hi != 0xA5).sum())
            first_hi = int(torch.nonzero(hi != 0xA5)[0]) if nhi else -1
            print(f"[{tag}] GUARD VIOLATED: {site} shape {shape} {dtype}: {nlo} bytes before, {nhi} bytes after (first at +{first_hi})", flush=True)
    print(f"[{tag}] {len(live)} guarded allocations, {bad} violated", flush=True)
    live.clear()


cfgs = [dict(B=2, L=16, img=64), dict(B=3, L=37, img=96), dict(B=2, L=130, img=224)]
for c in cfgs:
    for lowp in (torch.bfloat16,):
        torch.manual_seed(100)
        g = torch.Generator().manual_seed(3)
        B, L = c["B"], c["L"]
        ids = torch.randint(1000, 30000, (B, L), generator=g); ids[:, 0] = 101
        batch = tuple(t.to(dev) for t in (ids, torch.ones(B, L, dtype=torch.long), torch.zeros(B, L, dtype=torch.long),
                                         torch.randint(0, 3, (B,), generator=g), torch.randn(B, 3, c["img"], c["img"], generator=g)))
        tc = TextConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        vc = VisionConfig(num_hidden_layers=2, image_size=c["img"], patch_size=32)
        model = M.UnimoModelF(default_args(DR_step=3), vc, tc).to(dev)
        model.set_compute_dtype(lowp).train()
        model.model.use_streams = False
        store = ParamStore(model, lowp)
        loss, _ = model(*batch)   # warm-up outside the guards (workspaces, caches)
        loss.backward()
        torch.cuda.synchronize()
        torch.empty, torch.empty_like, torch.zeros = p_empty, p_empty_like, p_zeros
        try:
            store.zero_grad()
            loss, _ = model(*batch)
            loss.backward()
            from d2r_amd.functional import wgrad_join
            wgrad_join()
        finally:
            torch.empty, torch.empty_like, torch.zeros = _empty, _empty_like, _zeros
        check(f"B={B} L={L} img={c['img']} {str(lowp)[6:]}")
        del model, store

"""Is one training step a deterministic function of its inputs?  Runs the same fwd+bwd N times from identical weights (single
process, no collectives) and diffs the flat gradient buffers; with the two branch streams on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d2r_amd._lib as _L
if os.environ.get("PROBE_LIB"):  # experiment: another build of the library (e.g. routing.hip with packed fp32 math)
    _L.LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
    print("library under test:", _L.LIB_PATH, flush=True)
from d2r_amd import modules as M
from d2r_amd.config import TextConfig, VisionConfig, default_args
from d2r_amd.params import ParamStore

dev = torch.device("cuda:0")
from d2r_amd import functional as _F
_F.DETERMINISTIC = os.environ.get("D2R_DETERMINISTIC", "0") == "1"
M.COMPOSITE_ROUTING = os.environ.get("CR", "1") != "0"
M.COMPOSITE_LAYERS = os.environ.get("CL", "1") != "0"
N = int(os.environ.get("REPS", "8"))
g = torch.Generator().manual_seed(3)
C2 = os.environ.get("SHAPE") == "C2"  # the benchmark shape (batch 32, 128 text tokens, 197 image tokens, 12 + 12 encoder layers)
Bp, Lp, img, nl = (32, 128, 224, 12) if C2 else (2, 16, 64, 2)
ids = torch.randint(1000, 30000, (Bp, Lp), generator=g); ids[:, 0] = 101
batch = tuple(t.to(dev) for t in (ids, torch.ones(Bp, Lp, dtype=torch.long), torch.zeros(Bp, Lp, dtype=torch.long), torch.randint(0, 3, (Bp,), generator=g),
                                 torch.randn(Bp, 3, img, img, generator=g)))
torch.manual_seed(100)
tc = TextConfig(num_hidden_layers=nl, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
vc = VisionConfig(num_hidden_layers=nl, image_size=img, patch_size=16 if C2 else 32)
model = M.UnimoModelF(default_args(DR_step=3), vc, tc).to(dev)
model.set_compute_dtype(torch.bfloat16).train()
store = ParamStore(model, torch.bfloat16)
entries = [(n, o, k) for n, _, o, k, _ in store.entries]
bufs0 = {k: v.clone() for k, v in model.named_buffers()}
from d2r_amd import functional as F
captured = {}
_orig_mm = F.matmul_nt


def _mm(a, b):
    if a is b and a.requires_grad and a.dim() == 2 and a.shape[1] > 8:  # matmul_nt(paths, paths) of an interaction module
        key = f"d_paths_{a.shape[1]}_{len([k for k in captured if k.startswith('d_paths')]) % 2}"
        a.register_hook(lambda g, key=key: captured.__setitem__(key, g.detach().float().cpu().clone()))
    return _orig_mm(a, b)


if os.environ.get('HOOKS'):
    F.matmul_nt = _mm
    M.F.matmul_nt = _mm
cap_hist = []
snaps = []  # per step: [(tag, device clone of the forward arena / input gradients as the module's backward finds them)]
if os.environ.get("SNAP"):
    _orig_bwd2 = F._Interaction.backward

    def _bwd2(ctx, d_out, d_paths):
        cur = snaps[-1]
        tag = len(cur) // 3
        cur.append((f"arena{tag}", ctx.keep.clone()))
        cur.append((f"d_out{tag}", d_out.clone() if d_out is not None else None))
        cur.append((f"d_paths{tag}", d_paths.clone() if d_paths is not None else None))
        return _orig_bwd2(ctx, d_out, d_paths)

    F._Interaction.backward = staticmethod(_bwd2)
if os.environ.get("WAIT_ALL"):  # experiment: a module's backward first waits (on the GPU) for everything queued on the other compute streams
    _orig_bwd3 = F._Interaction.backward

    def _bwd3(ctx, d_out, d_paths):
        cur = torch.cuda.current_stream()
        which = os.environ.get("WAIT_WHICH", "all")
        cands = list(F._COMPUTE_STREAMS) if which in ("all", "side") else []
        cands += [torch.cuda.default_stream()] if which in ("all", "main") else []
        for st in cands:
            if st != cur:
                cur.wait_stream(st)
        res = _orig_bwd3(ctx, d_out, d_paths)
        if os.environ.get("WAIT_ALL") == "2":  # ... and the others wait for it
            for st in list(F._COMPUTE_STREAMS):
                if st != cur:
                    st.wait_stream(cur)
        return res

    F._Interaction.backward = staticmethod(_bwd3)
held = []
if os.environ.get("HOLD"):  # experiment: nothing the module's backward touched is returned to the allocator before the step ends
    _orig_bwd4 = F._Interaction.backward
    _orig_empty = torch.empty

    def _bwd4(ctx, d_out, d_paths):
        held.extend([d_out, d_paths, ctx.keep] + list(ctx.saved_tensors))
        if os.environ.get("HOLD") == "2":  # ... including the scratch buffers allocated inside
            def _empty(*a, **k):
                t = _orig_empty(*a, **k)
                held.append(t)
                return t
            torch.empty = _empty
        try:
            res = _orig_bwd4(ctx, d_out, d_paths)
        finally:
            torch.empty = _orig_empty
        held.extend([r for r in res if r is not None])
        return res

    F._Interaction.backward = staticmethod(_bwd4)
if os.environ.get("RECORD_ALL"):  # experiment: every tensor entering / leaving an interaction module's backward is marked as used on all streams
    _orig_bwd = F._Interaction.backward

    def _bwd(ctx, d_out, d_paths):
        streams_ = [torch.cuda.current_stream()] + list(F._COMPUTE_STREAMS) + [torch.cuda.default_stream()]
        keep = [t for t in list(ctx.saved_tensors) + [d_out, d_paths, ctx.keep] if t is not None]
        res = _orig_bwd(ctx, d_out, d_paths)
        for t in keep + [r for r in res if r is not None]:
            for st in streams_:
                t.record_stream(st)
        return res

    F._Interaction.backward = staticmethod(_bwd)
if os.environ.get("REV_ON_T"):  # experiment: the image-branch module runs on the TEXT stream (both whole-module calls on one stream)
    inner = model.model
    _orig_rev = inner.Reversed_itr_module.forward

    def _rev(*a, **k):
        if inner._streams is None or not inner.use_streams:
            return _orig_rev(*a, **k)
        sT, sV = inner._streams
        sT.wait_stream(sV)
        with torch.cuda.stream(sT):
            out = _orig_rev(*a, **k)
        sV.wait_stream(sT)
        for t in (out[0][0], out[1]):
            t.record_stream(sV)
        return out

    inner.Reversed_itr_module.forward = _rev
for streams in (True,):
    model.model.use_streams = streams
    grads = []
    for rep in range(N):
        with torch.no_grad():
            for k, v in model.named_buffers():
                v.copy_(bufs0[k])
        store.zero_grad()
        snaps.append([])
        loss, _ = model(*batch)
        if os.environ.get("SYNC_FB"):
            torch.cuda.synchronize()
        loss.backward()
        from d2r_amd.functional import wgrad_join
        wgrad_join()
        torch.cuda.synchronize()
        if C2:  # 1.4 GB of gradients per repetition: compared on the device against repetition 0
            if rep == 0:
                ref_dev = store.flat_g.detach().clone()
            else:
                nd = int((store.flat_g != ref_dev).sum())
                print(f"C2 shape rep {rep}: {'identical' if nd == 0 else str(nd) + ' gradient elements differ'}", flush=True)
            continue
        grads.append(store.flat_g.detach().cpu().clone())
        held.clear()
        if os.environ.get("SNAP") and rep > 0:
            for (tag, a0), (_, a1) in zip(snaps[0] if streams else snaps[0], snaps[-1]):
                if a0 is not None and not torch.equal(a0, a1):
                    nz = torch.nonzero((a0.view(torch.uint8) != a1.view(torch.uint8)).flatten())
                    print(f"streams={streams} rep {rep}: SNAPSHOT {tag} differs in {nz.numel()} bytes of {a0.numel() * a0.element_size()}, first {int(nz[0])} last {int(nz[-1])}", flush=True)
            snaps[-1] = None if rep > 0 else snaps[-1]
        cap_hist.append(dict(captured))
        captured.clear()
        if rep > 0:
            for k, v in cap_hist[0].items():
                if k in cap_hist[rep] and not torch.equal(v, cap_hist[rep][k]):
                    dd = (v - cap_hist[rep][k]).abs()
                    print(f"streams={streams} rep {rep}: INPUT GRADIENT {k} differs: cols {sorted(set(torch.nonzero(dd)[:, 1].tolist()))} max {float(dd.max()):.2e}", flush=True)
        aux = model.last_aux
        fw = getattr(sys.modules[__name__], "_fw", None)
        cur = [float(loss)] + [t.detach().float().cpu().clone() for t in (aux["emb_text"], aux["emb_image"], aux["sim_paths"], aux["rev_sim_paths"])]
        if rep == 0:
            fw0 = cur
        else:
            print(f"streams={streams} rep {rep}: forward loss equal {cur[0] == fw0[0]}, emb_text {torch.equal(cur[1], fw0[1])}, emb_image "
                  f"{torch.equal(cur[2], fw0[2])}, sim {torch.equal(cur[3], fw0[3])}, rev_sim {torch.equal(cur[4], fw0[4])}", flush=True)
    ref = grads[0] if grads else None
    for rep in range(1, N if grads else 0):
        d = grads[rep] != ref
        if bool(d.any()):
            cs = torch.cat([torch.zeros(1, dtype=torch.int64), d.to(torch.int64).cumsum(0)])
            names = [(n.replace("model.", ""), int(cs[o + k] - cs[o])) for n, o, k in entries if int(cs[o + k] - cs[o])]
            if os.environ.get("NAMES"):
                det = [(n, c, f"{float((grads[rep][o:o + k] - ref[o:o + k]).abs().max()):.1e}") for (n, c), (_, o, k) in
                       zip(names, [(n2, o2, k2) for n2, o2, k2 in entries if int(cs[o2 + k2] - cs[o2])]) if "itr_l1" in n or "itr_l2" in n]
                print("   l1/l2 detail:", det, flush=True)
            groups = {}
            for n, c in names:
                key = ".".join(n.split(".")[:3])
                groups[key] = groups.get(key, 0) + 1
            allg = {}
            for n, o, k in entries:
                key = ".".join(n.replace("model.", "").split(".")[:3])
                allg[key] = allg.get(key, 0) + 1
            print(f"streams={streams} rep {rep}: {int(d.sum())} elements differ in {len(names)} tensors, max |diff| "
                  f"{float((grads[rep] - ref).abs().max()):.3e}; differing tensors per group (of total): "
                  f"{[(k, v, allg[k]) for k, v in groups.items() if not k.startswith('block')]}", flush=True)
        else:
            print(f"streams={streams} rep {rep}: identical", flush=True)

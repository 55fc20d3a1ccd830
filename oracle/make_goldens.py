"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden-vector generator (run in the build container only).

    python -m oracle.make_goldens [case ...]          # writes tests/golden/*.npz

For every case in ``oracle/golden_cases.py`` this script
  1. builds the REAL reference module (imported from /root/reference, see ref_loader.py),
  2. loads name-keyed seeded weights (``d2r_oracle.seeded_state_dict``) with ``load_state_dict(strict=True)``
     — which also proves ``d2r_oracle.param_spec`` lists exactly the reference's keys and shapes,
  3. runs the reference forward + backward on seeded synthetic inputs TWICE: in fp64 (the "truth") and
     in fp32 (what the reference ships); the difference is stored as ``noise/*`` — the reference's own
     rounding noise, which is large for gradients because of the near-one-hot
     ``softmax(100*s/sqrt(768))`` chains (SURVEY.md section 7 "Hard parts"),
  4. runs the functional restatement ``oracle/d2r_oracle.py`` in fp64 and ASSERTS agreement with the
     fp64 reference to ~1e-9 (this pins the oracle's maths, gradients included), and in fp32 against the
     fp32 reference within the noise,
  5. stores the reference's outputs (data only — inputs, outputs, gradients) under tests/golden/.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import d2r_oracle as O
from . import ref_loader as R
from .golden_cases import MODEL_CASES, ROUTING_CASES, ModelCase, RoutingCase

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FULL_GRAD_KEYS_ROUTING = [
    "dynamic_itr_l0.ric.router.mlp.2.weight", "dynamic_itr_l0.glac.router.mlp.2.bias",
    "dynamic_itr_l0.glac.SAF_module.attn_sim_w.weight", "dynamic_itr_l0.glac.SAF_module.bn.weight",
    "dynamic_itr_l0.glac.SAF_module.bn.bias", "dynamic_itr_l2.cmrc.router.mlp.2.weight",
    "dynamic_itr_l2.gesc.fc_mlp.2.bias", "dynamic_itr_l1.0.imrc.sa.att_layer.linears.0.bias",
]
FULL_GRAD_KEYS_MODEL = [
    "fc.weight", "fc.bias", "model.itr_module.dynamic_itr_l0.ric.router.mlp.2.weight",
    "model.Reversed_itr_module.dynamic_itr_l2.glac.router.mlp.2.weight", "model.text_embeddings.LayerNorm.weight",
    "model.vision_embeddings.class_embedding", "model.vision_pre_layrnorm.bias",
    "model.text_embeddings.token_type_embeddings.weight", "model.block_fusion.linear_out.bias",
]


def _md(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max()) if a.numel() else 0.0


def _rel_l2(a, b, floor=0.0):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + floor + 1e-300))


def _check(name, a, b, atol, rtol=0.0, quiet=False):
    d = _md(a, b)
    scale = float(b.detach().abs().max()) if b.numel() else 0.0
    ok = d <= atol + rtol * scale
    if not quiet or not ok:
        print(f"    {name:<44s} max|Δ|={d:.3e} (scale {scale:.3e}) {'ok' if ok else 'MISMATCH'}")
    if not ok:
        raise AssertionError(f"oracle != reference for {name}: {d} (scale {scale})")


def _cast_sd(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}


def _leafify(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
            for k, v in sd.items()}


def _grad_report(ref64, ref32, or64, or32, dead_fn):
    """Compares gradient dicts; returns (names, norms64, noise_rel_l2) and asserts the oracle's maths."""
    names, norms, noise = [], [], []
    worst64 = 0.0
    gscale = max(float(g.abs().max()) for g in ref64.values() if g is not None) or 1.0
    gnorm = max(float(g.double().norm()) for g in ref64.values() if g is not None) or 1.0
    floor = 1e-6 * gnorm  # mathematically-zero gradients (e.g. a bias in front of BatchNorm) have no relative error
    for k, g64 in ref64.items():
        if g64 is None:
            assert dead_fn(k), f"{k} has no grad in the reference but is not listed dead"
            assert or64.get(k) is None or float(or64[k].abs().max()) == 0.0, f"{k}: reference has no grad, oracle has"
            continue
        assert not dead_fn(k), f"{k} listed dead but has a grad"
        assert or64.get(k) is not None, f"{k}: oracle has no grad"
        d = _md(or64[k], g64)
        sc = float(g64.abs().max())
        worst64 = max(worst64, d / (sc + 1e-9 * gscale))
        assert d <= 1e-8 * (sc + 1e-3 * gscale) + 1e-14, f"fp64 grad mismatch {k}: {d} (scale {sc})"
        names.append(k)
        norms.append(float(g64.double().norm()))
        noise.append(_rel_l2(ref32[k], g64, floor))
    o32 = sorted(_rel_l2(or32[k], ref64[k], floor) for k in names)
    n32 = sorted(noise)
    print(f"    {len(names)} parameter gradients: oracle64 vs ref64 worst rel {worst64:.1e}; "
          f"fp32 rel-L2 noise vs fp64 truth — reference median {n32[len(n32) // 2]:.1e} max {n32[-1]:.1e}, "
          f"oracle median {o32[len(o32) // 2]:.1e} max {o32[-1]:.1e}")
    return names, norms, noise


def routing_inputs(case: RoutingCase):
    g = torch.Generator().manual_seed(77 + case.seed)
    own = case.in_scale * torch.randn(case.B, case.Lq, 768, generator=g)
    other = case.in_scale * torch.randn(case.B, case.Lk, 768, generator=g)
    r_emb = torch.randn(case.B, case.Lq, 768, generator=g)
    r_sim = torch.randn(case.B, case.B, generator=g)
    return own, other, r_emb, r_sim


def _layer_names(dr):
    return ["dynamic_itr_l0"] + [f"dynamic_itr_l1.{i}" for i in range(dr - 2)] + ["dynamic_itr_l2"]


def _run_ref_routing(case, cfg, sd, dtype, own, other, r_emb, r_sim):
    ref, _ = R.build_reference_interaction(cfg, case.reversed_branch)
    ref.load_state_dict(sd, strict=True)
    ref = ref.to(dtype)
    ref.train(case.train)
    routers, layer_probs = {}, {}
    for n, m in ref.named_modules():
        if n.endswith(".router"):
            m.register_forward_hook(lambda mod, i, o, n=n: routers.__setitem__(n, o.detach().clone()))
        if n in _layer_names(case.DR_step):
            m.register_forward_hook(lambda mod, i, o, n=n: layer_probs.__setitem__(n, o[1].detach().clone()))
    own_r = own.detach().clone().to(dtype).requires_grad_(True)
    other_r = other.detach().clone().to(dtype).requires_grad_(True)
    text, image = (other_r, own_r) if case.reversed_branch else (own_r, other_r)
    emb_list, sim = ref(text, image)
    loss = (emb_list[0] * r_emb.to(dtype)).sum() + (sim * r_sim.to(dtype)).sum()
    loss.backward()
    res = dict(emb=emb_list[0].detach(), sim=sim.detach(), d_own=own_r.grad, d_other=other_r.grad, loss=loss.detach(),
               grads={k: p.grad for k, p in ref.named_parameters()},
               sd_after={k: v.detach().clone() for k, v in ref.state_dict().items()})
    for ln in _layer_names(case.DR_step):
        res["raw/" + ln] = torch.stack([routers[f"{ln}.{c}.router"] for c in O.CELL_ORDER], dim=2)
        res["probs/" + ln] = layer_probs[ln]
    return res


def _run_oracle_routing(case, sd, dtype, own, other, r_emb, r_sim):
    osd = _leafify(_cast_sd(sd, dtype))
    own_o = own.detach().clone().to(dtype).requires_grad_(True)
    other_o = other.detach().clone().to(dtype).requires_grad_(True)
    st, trace = O.BNState(case.train), {}
    emb, sim = O.interaction_module({"M." + k: v for k, v in osd.items()}, "M", own_o, other_o, case.DR_step, st, trace)
    loss = (emb * r_emb.to(dtype)).sum() + (sim * r_sim.to(dtype)).sum()
    loss.backward()
    return dict(emb=emb.detach(), sim=sim.detach(), d_own=own_o.grad, d_other=other_o.grad, trace=trace,
                grads={k: v.grad for k, v in osd.items() if v.is_floating_point() and v.requires_grad},
                bn={k[2:]: v for k, v in st.updates.items()})


def run_routing_case(case: RoutingCase):
    print(f"[routing] {case.name}")
    cfg = O.OracleConfig(DR_step=case.DR_step)
    sd = O.seeded_state_dict(cfg, seed=case.seed, router_bias=case.router_bias, spec=O.interaction_spec(cfg),
                             seed_prefix="rev." if case.reversed_branch else "fwd.")
    inp = routing_inputs(case)
    r64 = _run_ref_routing(case, cfg, sd, torch.float64, *inp)
    r32 = _run_ref_routing(case, cfg, sd, torch.float32, *inp)
    o64 = _run_oracle_routing(case, sd, torch.float64, *inp)
    o32 = _run_oracle_routing(case, sd, torch.float32, *inp)
    for k in ("emb", "sim", "d_own", "d_other"):
        _check(k + " (fp64)", o64[k], r64[k], 1e-10, 1e-10)
        _check(k + " (fp32 vs fp32 ref)", o32[k], r32[k], 3e-5, 3e-4)
    own, other, r_emb, r_sim = inp
    out = dict(own=own, other=other, r_emb=r_emb, r_sim=r_sim, emb=r64["emb"], sim_paths=r64["sim"],
               d_own=r64["d_own"], d_other=r64["d_other"], loss=r64["loss"])
    for k in ("emb", "sim", "d_own", "d_other"):
        out["noise/" + ("sim_paths" if k == "sim" else k)] = _md(r32[k], r64[k])
    for ln in _layer_names(case.DR_step):
        raw64, raw32 = r64["raw/" + ln], r32["raw/" + ln]
        _check(f"{ln}.raw_gates (fp64)", o64["trace"][f"M.{ln}.raw_gates"], raw64, 1e-12)
        _check(f"{ln}.probs (fp64)", o64["trace"][f"M.{ln}.probs"], r64["probs/" + ln], 1e-12)
        _check(f"{ln}.raw_gates (fp32)", o32["trace"][f"M.{ln}.raw_gates"], raw32, 2e-6, quiet=True)
        gm = (raw64 < 1e-4 / 6).double() if raw64.shape[1] == 1 else (raw64.sum(-1) < 1e-4).double()
        gm32 = (raw32 < 1e-4 / 6).double() if raw32.shape[1] == 1 else (raw32.sum(-1) < 1e-4).double()
        assert torch.equal(gm, gm32), f"{ln}: fp32 and fp64 reference disagree on a gate (borderline case)"
        assert torch.equal(raw64 > 0, raw32 > 0), f"{ln}: fp32 and fp64 reference disagree on open/closed"
        assert torch.equal(gm, o64["trace"][f"M.{ln}.gate_mask"].double()), f"gate mask mismatch in {ln}"
        assert torch.equal(raw64 > 0, o64["trace"][f"M.{ln}.raw_gates"] > 0), f"open/closed mismatch in {ln}"
        out[f"raw_gates/{ln}"] = raw64
        out[f"probs/{ln}"] = r64["probs/" + ln]
        out[f"gate_mask/{ln}"] = gm
        print(f"    {ln}: open paths {int((raw64 > 0).sum())}/{raw64.numel()}, skip gates set {int(gm.sum())}")
    names, norms, noise = _grad_report(r64["grads"], r32["grads"], o64["grads"], o32["grads"],
                                       lambda k: O.is_dead_param("model.itr_module." + k))
    out["grad_names"], out["grad_norms"], out["grad_noise"] = np.array(names), np.array(norms), np.array(noise)
    for k in FULL_GRAD_KEYS_ROUTING:
        if r64["grads"].get(k) is not None:
            out["grad/" + k] = r64["grads"][k]
    for k, v in o64["bn"].items():
        _check("bn " + k[-40:], v.double(), r64["sd_after"][k].double(), 1e-12, quiet=True)
        out["bn_after/" + k] = r64["sd_after"][k]
    _save(case.name, out)


def _run_ref_model(case, cfg, sd, dtype, batch):
    ids, mask, tt, labels, images = batch
    ref, _ = R.build_reference_model(cfg)
    ref.load_state_dict(sd, strict=True)
    ref = ref.to(dtype)
    ref.train(case.train)
    caught = {}
    ref.model.itr_module.register_forward_hook(lambda m, i, o: caught.__setitem__("t", (o[0][0].detach(), o[1].detach())))
    ref.model.Reversed_itr_module.register_forward_hook(
        lambda m, i, o: caught.__setitem__("v", (o[0][0].detach(), o[1].detach())))
    t0 = time.time()
    loss, logits = ref(ids, mask, tt, labels, images.to(dtype))
    loss.backward()
    print(f"    reference {str(dtype)[6:]} fwd+bwd {time.time() - t0:.1f}s")
    return dict(loss=loss.detach(), logits=logits.detach(), js_loss=(loss - F.cross_entropy(logits, labels)).detach(),
                emb_text=caught["t"][0], emb_image=caught["v"][0], sim_paths=caught["t"][1],
                rev_sim_paths=caught["v"][1], grads={k: p.grad for k, p in ref.named_parameters()},
                sd_after={k: v.detach().clone() for k, v in ref.state_dict().items()})


def _run_oracle_model(case, cfg, sd, dtype, batch):
    ids, mask, tt, labels, images = batch
    osd = _leafify(_cast_sd(sd, dtype))
    loss, logits, aux = O.forward(osd, cfg, ids, mask, tt, labels, images.to(dtype), train=case.train)
    loss.backward()
    return dict(loss=loss.detach(), logits=logits.detach(), js_loss=aux["js_loss"].detach(),
                emb_text=aux["emb_text"].detach(), emb_image=aux["emb_image"].detach(),
                sim_paths=aux["sim_paths"].detach(), rev_sim_paths=aux["rev_sim_paths"].detach(),
                grads={k: v.grad for k, v in osd.items() if v.is_floating_point() and v.requires_grad},
                bn=aux["bn_updates"])


OUT_KEYS = ("loss", "logits", "js_loss", "emb_text", "emb_image", "sim_paths", "rev_sim_paths")


def run_model_case(case: ModelCase):
    print(f"[model] {case.name}")
    cfg = case.cfg()
    sd = O.seeded_state_dict(cfg, seed=case.seed, router_bias=case.router_bias)
    batch = O.synthetic_batch(cfg, case.B, case.L, seed=case.seed)
    r64 = _run_ref_model(case, cfg, sd, torch.float64, batch)
    r32 = _run_ref_model(case, cfg, sd, torch.float32, batch)
    o64 = _run_oracle_model(case, cfg, sd, torch.float64, batch)
    o32 = _run_oracle_model(case, cfg, sd, torch.float32, batch)
    ids, mask, tt, labels, images = batch
    out = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels)
    if case.compact:  # regenerated by tests from the seed (d2r_oracle.synthetic_batch), checked against these sums
        out["images_sum"], out["images_abs_sum"] = images.double().sum(), images.double().abs().sum()
        out["images_probe"] = images[:, :, ::37, ::41].clone()
    else:
        out["images"] = images
    for k in OUT_KEYS:
        _check(k + " (fp64)", o64[k], r64[k], 1e-10, 1e-10)
        _check(k + " (fp32 vs fp32 ref)", o32[k], r32[k], 3e-5, 3e-4)
        if case.compact and k.startswith("emb_"):  # token 0 (what the poolers read) in full + the whole tensor's norm
            out[k + "_tok0"], out[k + "_norm"] = r64[k][:, 0].clone(), r64[k].double().norm()
        else:
            out[k] = r64[k]
        out["noise/" + k] = _md(r32[k], r64[k])
    print("    fp32 reference vs fp64 truth: " + ", ".join(f"{k} {out['noise/' + k]:.1e}" for k in OUT_KEYS))
    names, norms, noise = _grad_report(r64["grads"], r32["grads"], o64["grads"], o32["grads"], O.is_dead_param)
    out["grad_names"], out["grad_norms"], out["grad_noise"] = np.array(names), np.array(norms), np.array(noise)
    for k in FULL_GRAD_KEYS_MODEL:
        out["grad/" + k] = r64["grads"][k]
    for k, v in o64["bn"].items():
        _check("bn " + k[-40:], v.double(), r64["sd_after"][k].double(), 1e-12, quiet=True)
        out["bn_after/" + k] = r64["sd_after"][k]
    # weight-generator drift guard
    out["wsum/fc.weight"] = sd["fc.weight"].double().sum()
    out["wsum/word_emb"] = sd["model.text_embeddings.word_embeddings.weight"].double().sum()
    _save(case.name, out)


def _save(name, out):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    arrs = {}
    for k, v in out.items():
        a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if a.dtype == np.float64 and a.size > 4096:  # big activations: fp32 storage is plenty
            a = a.astype(np.float32)
        arrs[k] = a
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"    -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main(argv):
    if not R.reference_available():
        raise SystemExit("the reference is not mounted at /root/reference; goldens can only be made in the build container")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(argv[1:])
    for c in ROUTING_CASES:
        if not only or c.name in only:
            run_routing_case(c)
    for c in MODEL_CASES:
        if not only or c.name in only:
            run_model_case(c)


if __name__ == "__main__":
    main(sys.argv)

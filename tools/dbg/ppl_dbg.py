import sys, copy, types, torch
sys.path.insert(0, ".")
from tests import ppl_harness as ph, util
from kvquant_amd import calibrate, llama as kl
util.sync_oracle_freqs(10000.0)
dev = "cuda"
bits, n = 4, 96
model = ph.make_model(layers=1, vocab=4096, seed=0, device=dev, maxseqlen=n + 64, abits=bits, include_sparse=True)
g = torch.Generator().manual_seed(0)
ids = torch.randint(0, 4096, (1, n), generator=g).to(dev)
calib = torch.randint(0, 4096, (1, 2048), generator=g).to(dev)
quant = calibrate.calibrate_llama(model, calib, bits=bits)
# sim-deploy model
ms = copy.deepcopy(model)
at = ms.model.layers[0].self_attn
at.k_proj = ph.FakeQuantLinear(at.k_proj, quant["model.layers.0.self_attn.k_proj"], bits, True, 0.99, -1, False)
at.v_proj = ph.FakeQuantLinear(at.v_proj, quant["model.layers.0.self_attn.v_proj"], bits, False, 0.99, -1, False)
at.kvq_heads, at.kvq_hd, at.kvq_theta, at.kvq_ff = 32, 128, 10000.0, 0
at.forward = types.MethodType(ph._deploy_arith_forward, at)
outs = {}
def hook(name):
    def f(mod, inp, out): outs.setdefault(name, []).append(out[0].detach().float().clone())
    return f
at.register_forward_hook(hook("sim"))
with torch.no_grad(): ms(ids, use_cache=False)
mk = copy.deepcopy(model)
kl.patch_llama(mk); kl.load_quantizers(mk, quant)
mk.model.layers[0].self_attn.register_forward_hook(hook("ker"))
with torch.no_grad():
    for i in range(n): mk(ids[:, i:i+1], use_cache=False)
a = outs["sim"][0][0]                      # [T, hidden]
b = torch.cat([o[0] for o in outs["ker"]], 0)
err = (a - b).abs().amax(dim=1) / a.abs().amax(dim=1)
print("per-token rel err of layer-0 attention output (first 12):", [round(float(e), 5) for e in err[:12]])
print("max", float(err.max()), "median", float(err.median()), "argmax", int(err.argmax()))
# compare the caches' dequantised K against the sim K
kc = mk.model.layers[0].self_attn.kcache
from oracle import ckernels as ck
codes = ck.unpack_codes(bits, kc.kcache.cpu(), 4096, n)     # [L, C]
lut = kc.lookup_table.reshape(4096, 16).cpu()
kdeq = torch.gather(lut, 1, codes.long().t()).t()           # [L, C]
vals, idx = kc.outliers[:n].cpu(), kc.outlier_indices[:n].cpu().long()
kdeq.scatter_add_(1, idx, vals)
hs = []
def h2(mod, inp, out): hs.append(out.detach().float().reshape(-1, 4096).cpu())
hk = ms.model.layers[0].self_attn.k_proj.register_forward_hook(h2)
with torch.no_grad(): ms(ids, use_cache=False)
ksim = hs[0]
dk = (kdeq - ksim).abs()
print("K dequantised: kernel cache vs sim fake-quant: max abs diff %.4g, mean %.3g, frac > 1e-2: %.4g" % (float(dk.max()), float(dk.mean()), float((dk > 1e-2).float().mean())))
bad = (dk > 1e-2).nonzero()[:5]
for t, c in bad.tolist():
    print(" token", t, "channel", c, "kernel", float(kdeq[t, c]), "sim", float(ksim[t, c]), "thr", float(kc.outlier_threshold_lower[c]), float(kc.outlier_threshold_upper[c]))
# ---- V
vc = mk.model.layers[0].self_attn.vcache
vcodes = ck.unpack_codes(bits, vc.vcache.cpu(), 4096, n)
rows = vc.lookup_table[:n].cpu()                                  # [L, 16]
vdeq = torch.gather(rows, 1, vcodes.long())
vv, vi = vc.outliers[:n].cpu(), vc.outlier_indices[:n].cpu().long()
vdeq.scatter_add_(1, vi, vv)
hs.clear()
hk.remove()
hv = ms.model.layers[0].self_attn.v_proj.register_forward_hook(h2)
with torch.no_grad(): ms(ids, use_cache=False)
vsim = hs[0]
dv = (vdeq - vsim).abs()
print("V dequantised: kernel cache vs sim: max abs diff %.4g, mean %.3g, frac > 1e-2: %.4g" % (float(dv.max()), float(dv.mean()), float((dv > 1e-2).float().mean())))
print("per-token max V diff (first 16):", [round(float(x), 4) for x in dv.amax(dim=1)[:16]])
t = int(dv.amax(dim=1).argmax()); c = int(dv[t].argmax())
print(" worst token", t, "channel", c, "kernel", float(vdeq[t, c]), "sim", float(vsim[t, c]), "row", rows[t].tolist())

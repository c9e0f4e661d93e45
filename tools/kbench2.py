#!/usr/bin/env python
"""Per-launch timing of the DECODE variants of the hot-path kernels on one layer (development tool): decode
prologue, score kernel (query-premultiplied tables, token-contiguous outlier mirror, fused softmax partials),
softmax finish, p.V.  Two layer-sized caches alternate so that nothing is served from the 256 MB Infinity Cache.
usage: [KVQ_LIB=tools/abl/libkvq_X.so] python tools/kbench2.py [bits] [L ...]"""
import math
import os
import sys

import torch

sys.path.insert(0, ".")
if os.environ.get("KVQ_LIB"):
    import kvquant_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["KVQ_LIB"])
from kvquant_amd import ops  # noqa: E402

H, HD, C = 32, 128, 4096


def run(bits, L, iters=int(os.environ.get("KB_ITERS", "100")), nrot=2):
    n = 2 ** bits
    W = HD // 32 * bits
    max_len = (L + 64 + 63) // 64 * 64
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    caches = []
    for _ in range(nrot):
        d = {}
        d["k"] = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
        d["v"] = torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
        d["rows"] = torch.randn(max_len, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
        for nm in ("k", "v"):
            d[nm + "vals"] = torch.randn(max_len, 42, device=dev, generator=g)
            d[nm + "idx"] = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
        d["kvals_t"] = d["kvals"].t().contiguous()
        d["kidx_t"] = d["kidx"].t().contiguous()
        caches.append(d)
    lut = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
    q = torch.randn(1, H, HD, device=dev, generator=g)
    inv = 1.0 / math.sqrt(HD)
    s = torch.zeros(1, H, L, device=dev)
    out = torch.zeros(1, H, HD, device=dev)
    # tables of q into the score workspace
    ops.score_k(bits, q, caches[0]["k"], torch.zeros(1, H, 1, device=dev), lut, 1, 10000.0, 0, accumulate=False)
    ws = ops._workspace(dev, ops._L().kvq_score_k_workspace_bytes(bits, 1, H), slot="score")
    n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
    state = {}

    def kcall(i):
        d = caches[i % nrot]
        state["parts"] = ops.score_k_prepared_softmax(bits, d["k"], s, lut, L, 10000.0, 0, ws, d["kvals"], d["kidx"], inv,
                                                      n_parts, d["kvals_t"], d["kidx_t"])

    def fcall(i):
        state["p"], _ = ops.softmax_finish(s[0], state["parts"], n_parts, inv)

    def vcall(i):
        d = caches[i % nrot]
        if os.environ.get("KB_NOSPARSE"):
            ops.mix_v(bits, state["p"].unsqueeze(0), d["v"], out, d["rows"], L, None, None, accumulate=False)
        else:
            ops.mix_v(bits, state["p"].unsqueeze(0), d["v"], out, d["rows"], L, d["vvals"], d["vidx"], accumulate=False)

    def vfcall(i):
        d = caches[i % nrot]
        ops.mix_v_softmax(bits, s, state["parts"], n_parts, inv, d["v"], out, d["rows"], L, d["vvals"], d["vidx"])

    res = {}
    only = os.environ.get("KB_ONLY", "").split(",") if os.environ.get("KB_ONLY") else None
    for nm, fn, bpt in (("score_k", kcall, C * bits // 8 + 336 + 128), ("softmax_finish", fcall, 256),
                        ("mix_v", vcall, C * bits // 8 + 336 + 4 * n + 128),
                        ("mix_v_softmax", vfcall, C * bits // 8 + 336 + 4 * n + 128)):
        if only is not None and nm not in only:
            if nm == "score_k" or (nm == "softmax_finish" and "mix_v" in only):
                fn(0)                       # (later kernels read what it leaves behind)
            continue
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        # every launch between its own pair of events: the median is robust against clock / power transients
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i in range(iters):
            evs[i][0].record()
            fn(i)
            evs[i][1].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1000 for a, b in evs)
        us = ts[len(ts) // 2]
        res[nm] = us
        print("%s bits=%d L=%7d %-15s median %8.1f us (min %.1f, p90 %.1f)  %7.1f GB/s (%.1f%% of 8 TB/s)"
              % (os.environ.get("KVQ_LIB", "libkvq").split("/")[-1], bits, L, nm, us, ts[0], ts[int(len(ts) * 0.9)],
                 L * bpt / us / 1e3, L * bpt / us / 1e3 / 80), flush=True)
    return res


if __name__ == "__main__":
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for L in [int(a) for a in sys.argv[2:]] or [131072 + 77]:
        run(bits, L)

#!/usr/bin/env python
"""The reference's OWN kernels (oracle/_ref: its CUDA extension built for gfx950, see oracle/build_ref.py) timed on
the MI355X next to libkvq on identical inputs -- the per-op numbers BASELINE.md could not quote from the paper.
Development / documentation tool: imports the test-only reference build, so it is NOT part of bench.py.
usage: python tools/ref_bench.py [bits] [L ...]      -> one JSON line per (op, L)"""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from kvquant_amd import ops, quant_cuda  # noqa: E402
from oracle import build_ref  # noqa: E402

H, HD, C = 32, 128, 4096


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


def main():
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    Ls = [int(a) for a in sys.argv[2:]] or [4096, 32768, 131072]
    ref = build_ref.load()
    dev = torch.device("cuda:0")
    n = 2 ** bits
    W = HD // 32 * bits
    for L in Ls:
        max_len = (L + 64) // 64 * 64
        g = torch.Generator(device=dev).manual_seed(0)
        mats = [torch.randint(-2 ** 31, 2 ** 31 - 1, (H, W, max_len), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
                for _ in range(3)]
        lut = torch.randn(H, HD, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
        rows = torch.randn(max_len, n, device=dev, generator=g).sort(dim=-1).values.contiguous()
        vals = torch.randn(max_len, 42, device=dev, generator=g)
        idx = torch.sort(torch.randint(0, C, (max_len, 42), device=dev, generator=g), dim=-1).values.to(torch.int32)
        q = torch.randn(1, H, HD, device=dev, generator=g)
        p = torch.softmax(torch.randn(1, H, L, device=dev, generator=g), dim=-1).contiguous()
        mk = torch.zeros(1, H, L, device=dev)
        mv = torch.zeros(1, H, HD, device=dev)
        kname = "vecquant%dmatmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt2" % bits
        vname = "vecquant%dmatmul_nuq_perchannel_transposed_mha_batched_fused_opt2" % bits
        it = [0]

        def ref_k():
            it[0] += 1
            mk.zero_()
            getattr(ref, kname)(q, mats[it[0] % 3], mk, lut, L, vals, idx, 10000.0, 0)

        def our_k():
            it[0] += 1
            ops.score_k(bits, q, mats[it[0] % 3], mk, lut, L, 10000.0, 0, vals, idx, accumulate=False)

        # the module swap's call in the reference glue's decode pattern: one in-place row write per array
        # (ML:748-749), a fresh zeroed `mul` (ML:778), then the score call at a length one larger -- the shadow mirror
        # of kvquant_amd.quant_cuda takes one row per step
        Lq = [L - 40]
        newv, newi = vals[L - 1].clone(), idx[L - 1].clone()

        def qc_k():
            it[0] += 1
            vals[Lq[0]] = newv
            idx[Lq[0]] = newi
            Lq[0] += 1
            m = torch.zeros(1, H, Lq[0], device=dev)
            getattr(quant_cuda, kname)(q, mats[it[0] % 3], m, lut, Lq[0], vals, idx, 10000.0, 0)

        def ref_qc_k():      # the same pattern through the reference's kernel
            it[0] += 1
            vals[Lq[0]] = newv
            idx[Lq[0]] = newi
            Lq[0] += 1
            m = torch.zeros(1, H, Lq[0], device=dev)
            getattr(ref, kname)(q, mats[it[0] % 3], m, lut, Lq[0], vals, idx, 10000.0, 0)

        def ref_v():
            it[0] += 1
            mv.zero_()
            getattr(ref, vname)(p, mats[it[0] % 3], mv, rows, L, vals, idx)

        def our_v():
            it[0] += 1
            ops.mix_v(bits, p, mats[it[0] % 3], mv, rows, L, vals, idx, accumulate=False)
        iters = 5 if L > 200000 else 10
        if L > 64:
            quant_cuda.shadow_invalidate()
            before = dict(quant_cuda.shadow_stats)
            tq = timeit(qc_k, iters)
            did = {k: quant_cuda.shadow_stats[k] - before[k] for k in before}
            Lq[0] = L - 40
            tr = timeit(ref_qc_k, iters)
            Lq[0] = L - 40
            quant_cuda.QC_MIRROR = False
            trow = timeit(qc_k, iters)
            quant_cuda.QC_MIRROR = True
            print(json.dumps({"op": "q.K^T + sparse through quant_cuda's _opt2, decode pattern (row write + zeroed mul + call)",
                              "bits": bits, "L": L, "reference_kernel_us": round(tr, 1), "libkvq_us": round(tq, 1),
                              "libkvq_row_kernel_us": round(trow, 1), "shadow_mirror_calls": did,
                              "speedup": round(tr / tq, 2)}), flush=True)
            vt, ixt = vals.t().contiguous(), idx.t().contiguous()

            def our_km():
                it[0] += 1
                ops.score_k_mirror(bits, q, mats[it[0] % 3], mk, lut, L, 10000.0, 0, vt, ixt, accumulate=False)
            print(json.dumps({"op": "q.K^T + sparse, kvq_score_k_mirror alone (tables + kernel)", "bits": bits, "L": L,
                              "libkvq_us": round(timeit(our_km, iters), 1)}), flush=True)
            del vt, ixt
        for op, rf, of, bpt in (("q.K^T + sparse (legacy row layout)", ref_k, our_k, C * bits // 8 + 336 + 128),
                                ("p.V + sparse", ref_v, our_v, C * bits // 8 + 336 + 4 * n + 128)):
            tr, to = timeit(rf, iters), timeit(of, iters)
            print(json.dumps({"op": op, "bits": bits, "L": L, "reference_kernel_us": round(tr, 1), "libkvq_us": round(to, 1),
                              "speedup": round(tr / to, 2), "reference_GBps": round(L * bpt / tr / 1e3, 1),
                              "libkvq_GBps": round(L * bpt / to / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()

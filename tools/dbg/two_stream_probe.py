"""development probe (round 5): do the step's appends hide behind q.K^T when they run on a SECOND STREAM?
serial: prologue (K append | V append | tables) -> q.K^T -> p.V          (the round-4 launch order, five ctypes calls)
forked: tables -> [event] -> q.K^T ... [join] -> p.V on the main stream;  K append | V append on a side stream
Timing only (the forked variant scores the tokens cached before the step; the new token's own score is a few us of
work that the merge would absorb).  usage: python tools/dbg/two_stream_probe.py [ctx] [bits] [layers]"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from kvquant_amd import ops  # noqa: E402

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_layers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
H, HD, C = bench.H, bench.HD, bench.C
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(7)
steps = 6
max_len = (ctx + 2 * steps + 64 + 63) // 64 * 64
layers = []
for i in range(n_layers):
    lay = bench.Layer(bits, max_len, gen, dev, 0)
    lay.fill(ctx, gen, dev)
    layers.append(lay)
ks, vs = bench.synth_tokens(steps * 2, layers[0].scale, layers[0].shift, gen, dev)
qs = torch.randn(steps * 2, H, HD, generator=gen, device=dev).half()
inv = 1.0 / math.sqrt(HD)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def prologue(lay, q, k, v, col):
    kc, vc = lay.k, lay.v
    return ops.decode_prologue(bits, kc.kcache, kc.lookup_table, kc.lookup_table, k, kc.outlier_threshold_lower,
                               kc.outlier_threshold_upper, kc.outliers, kc.outlier_indices, col, vc.vcache,
                               vc.lookup_table, vc.lut, v, vc.outliers, vc.outlier_indices, col, q,
                               kc.num_outliers // 2, kc.outliers_t, kc.outlier_indices_t, kc.lut_ends, None, vc.vnorm_args())


def attend(lay, ws, L, out):
    kc, vc = lay.k, lay.v
    scores = torch.empty((1, H, L), dtype=torch.float32, device=dev)
    n_parts = ops._L().kvq_score_k_softmax_parts(bits, L, 1)
    parts = ops.score_k_prepared_softmax(bits, kc.kcache, scores, kc.lookup_table, L, kc.rope_theta, 0, ws, kc.outliers,
                                         kc.outlier_indices, inv, n_parts, kc.outliers_t, kc.outlier_indices_t)
    return scores, parts, n_parts


def mix(lay, scores, parts, n_parts, L, out):
    vc = lay.v
    ops.mix_v_softmax(bits, scores, parts, n_parts, inv, vc.vcache, out, vc.mix_table(), L, vc.outliers, vc.outlier_indices)


def run(forked, step0):
    out = torch.empty((1, H, HD), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        col = ctx + step0 + s
        for lay in layers:
            q, k, v = qs[step0 + s], ks[step0 + s], vs[step0 + s]
            if not forked:
                ws = prologue(lay, q, k, v, col)
                scores, parts, n_parts = attend(lay, ws, col + 1, out)
                mix(lay, scores, parts, n_parts, col + 1, out)
            else:
                ws = ops.score_k_tables(bits, q, lay.k.lookup_table, H)
                e1 = torch.cuda.Event()
                e1.record(main)
                side.wait_event(e1)
                with torch.cuda.stream(side):
                    kc, vc = lay.k, lay.v
                    ops.append_k_fused(bits, kc.kcache, kc.lookup_table, kc.lookup_table, k.float(), kc.outlier_threshold_lower,
                                       kc.outlier_threshold_upper, kc.outliers, kc.outlier_indices, kc.num_outliers // 2, col,
                                       kc.outliers_t, kc.outlier_indices_t)
                    ops.append_v_fused(bits, vc.vcache, vc.lookup_table, vc.lut, v.float(), vc.outliers, vc.outlier_indices,
                                       vc.num_outliers // 2, col, vc.vnorm_args())
                    e2 = torch.cuda.Event()
                    e2.record(side)
                scores, parts, n_parts = attend(lay, ws, col, out)      # (the tokens cached before the step)
                main.wait_event(e2)
                mix(lay, scores, parts, n_parts, col, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


# (k / v as fp32 for the single-token append entry points; converted outside the timed region)
ks, vs, qs = ks.float(), vs.float(), qs.float()
for name, forked in (("serial", False), ("forked", True), ("serial", False), ("forked", True)):
    step0 = 0 if name == "serial" else steps
    for lay in layers:      # rewind the caches to ctx tokens
        lay.k.klen = ctx
        lay.v.vlen = ctx
    ms = run(forked, step0)
    print("ctx=%d bits=%d layers=%d %s: %.3f ms/step (%.1f us per layer)" % (ctx, bits, n_layers, name, ms, ms * 1e3 / n_layers))

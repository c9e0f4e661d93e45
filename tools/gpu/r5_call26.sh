#!/bin/bash
# round 5, call 26: p.V with 256-lane workgroups (4 per CU) against 512 (2 per CU) -- dense part only (no outlier rows), non-fused
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r05_m
for rep in 1 2; do
for lib in default v256; do
  if [ $lib = default ]; then unset KVQ_LIB; else export KVQ_LIB=tools/abl/libkvq_$lib.so; fi
  for bits in 4 3; do
  KB_NOSPARSE=1 KB_ONLY=mix_v KB_ITERS=60 timeout 300 python tools/kbench2.py $bits 131072 2>&1 | grep "mix_v " >> ${O}_v_nt.txt
  done
done; done
cat ${O}_v_nt.txt
# correctness of the variant's dense path
KVQ_LIB=tools/abl/libkvq_v256.so python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import kvquant_amd._lib as _l
_l.LIB_PATH = os.path.abspath("tools/abl/libkvq_v256.so")
from kvquant_amd import ops
import ctypes
ref = ctypes.CDLL(os.path.abspath("kvquant_amd/libkvq.so"))
H, HD = 32, 128
for bits in (4, 3, 2):
    for L in (700, 40000):
        n = 2 ** bits; W = HD // 32 * bits; max_len = (L + 127) // 64 * 64
        g = torch.Generator(device="cuda").manual_seed(1)
        mat = torch.randint(-2**31, 2**31 - 1, (H, W, max_len), device="cuda", dtype=torch.int64, generator=g).to(torch.int32)
        rows = torch.randn(max_len, n, device="cuda", generator=g)
        p = torch.softmax(torch.randn(1, H, L, device="cuda", generator=g), -1).contiguous()
        out = torch.zeros(1, H, HD, device="cuda")
        ops.mix_v(bits, p, mat, out, rows, L, None, None, accumulate=False)
        # reference: dequantise in torch
        codes = torch.stack([((mat.view(H, HD // 32, bits, max_len)[:, :, :, :L].long() & 0xffffffff)) ], 0)[0]
        # unpack 32 codes of `bits` bits from `bits` words (little-endian bit stream)
        words = codes  # [H, G, bits, L]
        vals = torch.zeros(H, HD // 32, 32, L, device="cuda", dtype=torch.long)
        for c in range(32):
            b0 = c * bits; w0, o = b0 // 32, b0 % 32
            v = (words[:, :, w0] >> o)
            if o + bits > 32: v = v | (words[:, :, w0 + 1] << (32 - o))
            vals[:, :, c] = v & (n - 1)
        deq = rows[:L].t()[vals.reshape(H * HD, L), torch.arange(L, device="cuda")[None, :]] if False else torch.gather(rows[:L].t().contiguous(), 0, vals.reshape(H * HD, L))
        want = (deq.view(H, HD, L).double() * p[0].double()[:, None, :]).sum(-1)
        err = float((out[0].double() - want).abs().max() / want.abs().max())
        print("v256 dense bits %d L %d rel err %.2e" % (bits, L, err))
PY

#!/usr/bin/env python
"""BASELINE config 1 harness: PPL of the reference's simulated-quant path vs the kernel path on the same random-init
Llama (see tests/ppl_harness.py).  Prints one JSON line per configuration.
usage: python tools/ppl_delta.py [n_tokens]"""
import json
import sys

sys.path.insert(0, ".")
from tests import ppl_harness  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for kw in (dict(bits=4), dict(bits=4, n_prompt=n // 2), dict(bits=3, first_few_fp16=5), dict(bits=2, norm=True)):
    print(json.dumps(ppl_harness.run(n_tokens=n, **kw)), flush=True)

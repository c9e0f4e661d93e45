#!/usr/bin/env python
"""BASELINE config 1 harness: PPL of the reference's simulated-quant path vs the kernel path on the same Llama (7B head
shape, 2 layers; see tests/ppl_harness.py).  Prints one JSON line per configuration.
usage: python tools/ppl_delta.py [n_tokens] [train_steps] [repeats]
  train_steps > 0 (default 300): the model is first trained on a synthetic Markov stream whose ideal perplexity is 5.48
  (wikitext-2 on LLaMA-2-7B: 5.47), so that the deltas are perplexity differences of a model that predicts;
  train_steps = 0: the random-init proxy of rounds 1-3 (PPL ~ vocabulary size)."""
import json
import sys

sys.path.insert(0, ".")
from tests import ppl_harness  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # (every run draws fresh evaluation tokens from the stream)
vocab = 4096 if steps > 0 else 32000
for kw in reps * (dict(bits=4), dict(bits=4, n_prompt=n // 2), dict(bits=3, first_few_fp16=5), dict(bits=2, norm=True)):
    print(json.dumps(ppl_harness.run(n_tokens=n, vocab=vocab, train_steps=steps,
                                     log=lambda m: print(m, file=sys.stderr, flush=True), **kw)), flush=True)

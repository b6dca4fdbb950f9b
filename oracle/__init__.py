"""CPU oracle for the speech-prompted-LLM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path
(`llm-speech-summarization_amd/`) may import this package; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
only as the checker / reported baseline.

The oracle is a plain PyTorch-CPU fp32 restatement of the arithmetic that the
reference (wonjune-kang/llm-speech-summarization) delegates to the third-party
`transformers` package (pinned 4.47.0 in ref:requirements.txt:15, not vendored
under /root/reference).  It deliberately does NOT import `transformers` or the
reference: every function cites the reference / HF file:line it restates.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference classes themselves, run in
the build container by `oracle/gen_golden.py` and frozen under
`tests/golden/*.npz` (checked by `tests/test_oracle_golden.py`).
"""

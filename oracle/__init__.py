"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (pure PyTorch / pure Python loops, written from scratch) of the
one hot path of facebookresearch/metamorph that `metamorph_amd` accelerates:
SigLIP tower -> token-reduce + L2 norm -> mm_projector -> <image>/text splice ->
LLaMA decoder -> {lm_head + CE, vision_head + cosine}.

Rules (enforced by tests/test_host_logic.py::test_product_never_imports_oracle_or_reference):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything from this package;
  * nothing under `metamorph_amd/` imports it -- the product fails loudly when
    the HIP library is missing, it never falls back to this code.

Pinning: the reference has no tests or golden vectors of its own, so
the pin is `tests/golden/*.npz`, produced by `oracle/gen_golden.py`, which imports
the reference's own Python from /root/reference (plus pinned-elsewhere
`transformers`, installed 5.15.0 vs the reference's 4.45.0 pin -- same math, see
DESIGN.md) and records its outputs.  `tests/test_oracle_vs_golden.py` checks every
function here against those vectors.
"""

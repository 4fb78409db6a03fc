"""Import-only stand-in for `deepspeed` (TEST INFRASTRUCTURE, oracle/gen_golden.py only): the reference's `maybe_zero_3`
(train.py:116-127, metamorph_trainer.py:23-33) imports `deepspeed.zero` and `ZeroParamStatus` before testing `hasattr(param, "ds_id")`;
plain parameters take the `else` branch, so nothing here is ever executed."""
from . import zero  # noqa: F401

class GatheredParameters:            # only entered for ZeRO-3 partitioned parameters (never in the fixtures)
    def __init__(self, *a, **k):
        raise RuntimeError("deepspeed shim: no partitioned parameters exist in the golden fixtures")

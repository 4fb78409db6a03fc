class ZeroParamStatus:
    NOT_AVAILABLE = 0

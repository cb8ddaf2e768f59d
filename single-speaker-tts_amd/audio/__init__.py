"""Mirror of the reference's ``audio`` package for the inference path (conversion, synthesis,
features, io).  All array arithmetic runs in libsstts_hip.so."""
from .._hip import Engine

_default_engine = None


def default_engine():
    """Process-wide Engine (device 0, no weights) used by the module-level audio functions."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


def set_default_engine(engine):
    global _default_engine
    _default_engine = engine

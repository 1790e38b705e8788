"""grove_amd — MI355X-native implementation of GROVE's per-clip forward/backward hot path.

    from grove_amd import GROVEForCausalLM
"""
from .synthetic import FULL, TINY, GroveDims  # noqa: F401


def __getattr__(name):
    if name == "GROVEForCausalLM":
        from .model.GROVE import GROVEForCausalLM
        return GROVEForCausalLM
    raise AttributeError(name)

"""Seed of the device Philox streams (the reference seeds nothing: python `random` state)."""
_state = {"seed": 0x5EED0EA, "calls": 0}


def set_seed(seed):
    _state["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _state["calls"] = 0


def get_seed():
    return _state["seed"]


def next_call():
    """a fresh `step` counter value for API-level sampling calls (distinct streams per call)."""
    _state["calls"] += 1
    return 0x40000000 + (_state["calls"] & 0x3FFFFFFF)

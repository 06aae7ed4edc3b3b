from ._seed import set_seed  # noqa: F401

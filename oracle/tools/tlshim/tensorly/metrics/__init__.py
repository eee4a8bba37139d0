from . import factors  # noqa: F401

from . import strategies  # noqa: F401

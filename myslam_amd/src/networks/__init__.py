from . import decoders  # noqa: F401

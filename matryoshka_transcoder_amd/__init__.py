"""MI355X-native Matryoshka transcoder hot path (libmtc.so + host mirror of the reference API)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]

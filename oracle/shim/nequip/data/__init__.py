from . import AtomicDataDict  # noqa: F401

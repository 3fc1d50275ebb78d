from . import inference, group  # noqa: F401

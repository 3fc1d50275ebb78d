from . import pose_mobilenet  # noqa: F401

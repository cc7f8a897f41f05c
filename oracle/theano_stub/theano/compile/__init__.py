from . import sharedvalue                    # noqa: F401

from .._lazy import SharedVariable, shared   # noqa: F401

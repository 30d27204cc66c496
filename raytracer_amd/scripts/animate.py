"""Per-frame / per-time animation hooks (reference ``scripts/animate.py:21-85``).

A hook is just ``fun(obj, t)``; the scene evaluates its hooks in list order
(``Scene.animate_frame``), which is what makes a frame a pure function of its index and
lets frames shard across GPUs with no exchange (multi-GPU).
"""
from __future__ import annotations

from typing import Callable, Generic, TypeVar

T = TypeVar("T")


class _Hook(Generic[T]):
    def __init__(self, obj: T, fun: Callable[[T, float], T]) -> None:
        self.obj = obj
        self.fun = fun

    def evaluate(self, t):
        return self.fun(self.obj, t)


class FrameAnimation(_Hook):
    """Hook keyed by integer frame index."""


class Animation(_Hook):
    """Hook keyed by time in seconds."""

    def reset(self):
        raise NotImplementedError

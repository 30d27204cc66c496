"""The reference's alias for "the scene module in use" (``scripts/scenes/loaded_scene.py``:
``from .dragon_scene import *``).  Here the dragon scene is built lazily on first access of ``scene``,
so the alias forwards attribute look-ups instead of star-importing."""
from . import dragon_scene as _module


def __getattr__(name):
    return getattr(_module, name)

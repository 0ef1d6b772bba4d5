"""The modelling vocabulary of the reference (`from codecad_amd.shapes import *`, reference
shapes/__init__.py:19-109): constructor functions live next to the classes they build -- primitives.py,
combinators.py, polygons2d.py -- and are gathered here; `unsafe` and `gears` are submodules."""
from . import simple2d, simple3d  # noqa: F401  (class names under the reference's module names)
from . import unsafe  # noqa: F401
from . import gears  # noqa: F401
from .base import TapeShape  # noqa: F401
from .primitives import (rectangle, circle, half_plane, regular_polygon2d, capsule, box, sphere,  # noqa: F401
                         cylinder, half_space)
from .combinators import union, intersection  # noqa: F401
from .polygons2d import polygon2d, polygon2d_builder  # noqa: F401

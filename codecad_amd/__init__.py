"""codecad_amd: MI355X-native SDF evaluation behind the bluecube/codecad modelling surface.

    import codecad_amd as codecad
    s = codecad.shapes.sphere(130) & codecad.shapes.box(100)
    codecad.mass_properties(s, 1.0)

Public layout mirrors the reference package (reference codecad/__init__.py:1-11) for the
hot path only: `shapes`, `util`, `nodes`, `hip_util` (in place of `cl_util`), `grid_eval`,
`subdivision`, `mass_properties`.  Renderers, assemblies and the CLI are out of scope
(DESIGN.md).  Importing the package does not touch the GPU; the first kernel launch does,
and raises if the HIP library or a device is missing -- there is no CPU fallback.
"""
from . import util  # noqa: F401
from . import nodes  # noqa: F401
from . import shapes  # noqa: F401
from . import hip_util  # noqa: F401
from . import grid_eval  # noqa: F401
from . import subdivision  # noqa: F401
from .mass_properties import mass_properties, MassProperties  # noqa: F401
from . import examples  # noqa: F401
from . import rendering  # noqa: F401

__all__ = ["util", "nodes", "shapes", "hip_util", "grid_eval", "subdivision", "mass_properties",
           "MassProperties", "examples"]

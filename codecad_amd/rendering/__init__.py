"""Renderers built on the same evaluate() (reference codecad/rendering/): the sphere-tracing ray
caster for 3D shapes and the inside/outside bitmap for 2D shapes, as one `render_image` (SURVEY.md section
8(f) rank 3), and the 2D contouring `polygon2d.polygon` with its SVG writer (rank 4).  The rest of the
reference's rendering package (matplotlib viewers, animations, the CLI dispatch) is out of scope."""
from . import ray_caster, bitmap, image, polygon2d, svg  # noqa: F401
from .image import render_image, render_pil_image  # noqa: F401
from .svg import render_svg  # noqa: F401

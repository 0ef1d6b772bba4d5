"""Renderers built on the same evaluate() (reference codecad/rendering/): the sphere-tracing ray
caster for 3D shapes and the inside/outside bitmap for 2D shapes, as one `render_image`.
SURVEY.md section 8(f) rank 3; everything else of the reference's rendering package (mesh/STL,
SVG contouring, matplotlib viewers, the CLI) is out of scope."""
from . import ray_caster, bitmap, image  # noqa: F401
from .image import render_image, render_pil_image  # noqa: F401

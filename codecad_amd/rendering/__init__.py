"""Renderers built on the same evaluate() (reference codecad/rendering/): the sphere-tracing ray
caster for 3D shapes and the inside/outside bitmap for 2D shapes, as one `render_image` (SURVEY.md section
8(f) rank 3), the 2D contouring `polygon2d.polygon` with its SVG writer (rank 4), and the leaf-block
consumer `mesh.triangular_mesh` (marching cubes on the device) with its STL writer (rank 2).  The rest of
the reference's rendering package (matplotlib viewers, animations, the CLI dispatch) is out of scope."""
from . import ray_caster, pictures, polygon2d, mesh, stl_renderer  # noqa: F401
from .stl_renderer import render_stl  # noqa: F401
from .pictures import render_image, render_pil_image, render_pixels  # noqa: F401
from .polygon2d import render_svg  # noqa: F401

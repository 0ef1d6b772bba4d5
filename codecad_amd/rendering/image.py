"""`render_image(shape, filename)`: ray caster for 3D, bitmap for 2D (reference rendering/image.py)."""
from . import ray_caster, bitmap


def render_pixels(obj, size=(1024, 768), view_angle=None):
    if obj.dimension() == 2:
        return bitmap.render(obj, size)
    camera = ray_caster.get_camera_params(obj.bounding_box(), size, view_angle)
    return ray_caster.render(obj, *camera, size=size)


def render_pil_image(obj, size=(1024, 768), view_angle=None):
    import PIL.Image
    return PIL.Image.fromarray(render_pixels(obj, size, view_angle))


def render_image(obj, filename, size=(1024, 768), view_angle=None):
    render_pil_image(obj, size, view_angle).save(filename)

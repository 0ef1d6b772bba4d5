import sys
sys.path.insert(0, '.')
import codecad_amd as codecad
s = codecad.examples.sponge(4)
print(codecad.mass_properties(s, 1 / 81, grid_size=9).volume)
tape, dims, blocks = codecad.subdivision.subdivision(s, 1 / 512, grid_size=16)
print(dims, len(blocks))
codecad.rendering.render_image(s.rotated((1, 2, 3), 40), "/tmp/sponge.png")
print(codecad.rendering.render_stl(codecad.shapes.sphere(20) - codecad.shapes.cylinder(h=30, d=8), "/tmp/part.stl"))
codecad.rendering.render_svg(codecad.shapes.regular_polygon2d(6, 10).offset(1), "/tmp/hexagon.svg")
print(open("/tmp/hexagon.svg").read()[:200])
t = codecad.nodes.make_program_buffer(s).specialize()
print(t.specialized)

"""The shapes the golden fixtures were generated for, built with THIS package's API.

Data only: the same parameter values as the reference's own test fixtures (reference
tests/data.py:7-96, tests/test_mass_properties.py:16-96, tests/test_subdivision.py:110-141,
tests/test_polygons2d.py:9-25) so that tests/golden/ref_tapes.json (tapes the reference's
compiler produced for them) can be compared with what our compiler produces.
"""
import math

import numpy

from codecad_amd import util
from codecad_amd.shapes import (box, sphere, cylinder, circle, rectangle, half_space, union, polygon2d,
                                regular_polygon2d, gears)
from codecad_amd.examples import sponge, csg_example, sphere_plus_box

valid_polygon2d = {
    "triangle": [(0, 0), (3, 0), (3, 2)],
    "non_convex": [(0, 0), (3, 0), (3, 1), (2, 2), (3, 3), (0, 3)],
    "collinear_consecutive_edges": [(0, 0), (2, 0), (4, 0), (4, 3)],
    "collinear_non_consecutive_edges": [(0, 0), (3, 0), (3, 1), (2, 2), (3, 3), (3, 4), (0, 4)],
    "square": [(0, 0), (5, 0), (5, 5), (0, 5)],
    "parallel_same_direction_edges": [(0, 0), (6, -1), (5, 5), (5, 0), (0, 5)],
}

invalid_polygon2d = {
    "edge_crossing": [(0, 0), (3, 0), (0, 3), (3, 3)],
    "point_crossing": [(0, 0), (3, 0), (2.5, 2.5), (0, 3), (3, 3), (2.5, 2.5)],
    "repeated_point_on_collinear_edges": [(0, 0), (3, 0), (3, 2), (2, 1), (2, 3), (3, 2), (3, 4), (0, 4)],
    "collinear_edge_crossing": [(0, 0), (3, 0), (3, 3), (2, 2), (3, 1), (3, 4), (0, 4)],
    "repeated_point": [(0, 0), (4, 0), (4, 3), (0, 0), (1, 3), (0, 3)],
    "point_on_edge": [(0, 0), (4, 0), (4, 3), (2, 0), (0, 3)],
    "shared_edge": [(0, 0), (3, 0), (5, 0), (3, 0), (3, 3)],
    "shared_edge_part": [(0, 0), (5, 0), (3, 0), (3, 3)],
    "duplicate_point": [(0, 0), (2, 0), (2, 0), (4, 3)],
    "duplicate_point_at_start": [(0, 0), (3, 0), (3, 2), (0, 0)],
}


def bin_counter(n):
    blip = circle(d=0.75)
    blips, bits = [], 0
    while n > 0:
        if n & 1:
            blips.append(blip.translated_y(0.5 + bits))
        n //= 2
        bits += 1
    bits = max(bits, 1)
    base = rectangle(1, bits).translated_y(bits / 2) + circle(d=0.2).translated_x(0.5)
    return base - union(blips) if blips else base


nonconvex = polygon2d([(0, 0), (4, 6), (4, -2), (-4, -2), (-4, 6)])
csg_thing = (cylinder(h=5, d=2, symmetrical=False).rotated((1, 2, 3), 15) & sphere(d=3)) + \
    box(2).translated(.5, 0, -.5)
_m2 = rectangle(1, 4).translated_x(-0.5) + circle(r=1).translated_y(-1)
mirror_2d = union([_m2, _m2.mirrored_x().translated_x(5), _m2.mirrored_y().translated_y(5)])
_m3 = box(1, 4, 4).translated_x(-0.5) + sphere(r=1).translated(0, -1, -1)
mirror_3d = union([_m3, _m3.mirrored_x().translated_x(5), _m3.mirrored_y().translated_y(5),
                   _m3.mirrored_z().translated_z(5)])

shapes_2d = {
    "rectangle": rectangle(2, 4),
    "circle": circle(4),
    "nonconvex_offset_outside": nonconvex.offset(2),
    "nonconvex_offset_inside1": nonconvex.offset(-0.9),
    "nonconvex_offset_inside2": nonconvex.offset(-1.1),
    "nonconvex_shell1": nonconvex.shell(1),
    "nonconvex_shell2": nonconvex.shell(2.5),
    "gear": gears.InvoluteGear(20, 0.5),
    "mirror_2d": mirror_2d,
    "bin_counter_11": bin_counter(11),
    "regular_polygon3": regular_polygon2d(3),
    "symmetrical_xy": circle(d=2).translated(2, 1.75).symmetrical_x().symmetrical_y(),
    "rotated_pattern_2d": circle(1, 1).translated_x(2).rotated(270, 3),
}
shapes_2d.update(("polygon2d_" + k, polygon2d(v)) for k, v in valid_polygon2d.items())

shapes_3d = {
    "sphere": sphere(4),
    "box": box(2, 3, 5),
    "drunk_box": box(2, 3, 5).rotated((7, 11, 13), 17),
    "translated_cylinder": cylinder(d=3, h=5).translated(0, 1, -1),
    "csg_thing": csg_thing,
    "torus": circle(d=4).translated_x(3).revolved(),
    "empty_intersection": sphere().translated_x(-2) & sphere().translated_x(2),
    "nested_transformations": (box().translated_z(-2) + sphere().translated_x(2)).rotated_y(45).rotated_x(45),
    "mirror_3d": mirror_3d,
    "revolved_pentagon": regular_polygon2d(5).revolved(2),
    "extreme_twisted_revolve": rectangle(1, 0.1).revolved(2, 19 * 180),
    "symmetrical_xyz": sphere(d=2).translated(2, 2, 2).symmetrical_x().symmetrical_y().symmetrical_z(),
    "rotated_pattern_3d": box(1, 1, 1).translated_x(2).rotated((1, -1, 0), 270, 3),
}

drunk_box_matrix = util.Quaternion.from_degrees((7, 11, 13), 17).as_matrix()[:3, :3]

# name -> (shape, volume, centroid, inertia tensor or None): reference
# tests/test_mass_properties.py:16-96 (analytic values)
mass_property_cases = {
    "unit_box": (box(1), 1, (0, 0, 0), numpy.identity(3) * 2 / 12),
    "cylinder": (cylinder(h=2, r=4, symmetrical=False), math.pi * 32, (0, 0, 1),
                 numpy.diag([(3 * 4 ** 2 + 2 ** 2) / 6, (3 * 4 ** 2 + 2 ** 2) / 6, 4 ** 2]) * math.pi * 16),
    "sphere": (sphere(d=2), 4 * math.pi / 3, (0, 0, 0), numpy.identity(3) * (4 * math.pi / 3) * 2 / 5),
    "two_boxes": (box(2).translated(-15, 0, 0) + box(2).translated(15, 0, 0), 16, (0, 0, 0), None),
    "hemisphere": (sphere(r=2) - half_space(), 2 * math.pi * 2 ** 3 / 3, (0, -6 / 8, 0), None),
    "translated_sphere": (sphere(d=2).translated(10, 11, 7), 4 * math.pi / 3, (10, 11, 7), None),
    "translated_and_rotated_hemisphere": ((sphere(r=2) - half_space()).translated(2, 0, 0).rotated((1, 0, 0), 90),
                                          2 * math.pi * 2 ** 3 / 3, (2, 0, -6 / 8), None),
    "not_hammer": (box(4).translated(0, 0, 2) + box(2, 2, 9).translated(0, 0, -3.5), 96, (0, 0, 0),
                   numpy.diag([1120, 1120, 192])),
    "drunk_box": (box(2, 3, 5).rotated((7, 11, 13), 17), 2 * 3 * 5, (0, 0, 0),
                  drunk_box_matrix @ (numpy.diag([3 ** 2 + 5 ** 2, 2 ** 2 + 5 ** 2, 2 ** 2 + 3 ** 2]) * 2 * 3 * 5 / 12)
                  @ drunk_box_matrix.T),
}

_kat_res, _kat_g = 0.1, 8
kat_circle_diameter = _kat_g * (_kat_res * (_kat_g - 1)) - _kat_res

config_shapes = {"sphere_plus_box": sphere_plus_box(), "csg_example": csg_example()}
config_shapes.update(("sponge%d" % n, sponge(n)) for n in range(6))

all_named = {}
all_named.update(config_shapes)
all_named.update(shapes_2d)
all_named.update(shapes_3d)
all_named.update(("mp_" + k, v[0]) for k, v in mass_property_cases.items())
all_named["kat_box10"] = box(10)
all_named["kat_circle"] = circle(kat_circle_diameter)

# Rounded blends (union / intersection with r >= 0, reference shapes/common.cl:45-64): the one op
# through which a direction feeds a distance.  Not in the reference's test zoo; compiled by our
# compiler only (no golden tape), checked HIP-vs-oracle and for the blend's own properties.
rounded_shapes = {
    "rounded_union_3d": union([box(2), sphere(2.5).translated(1, 0, 0)], r=0.4),
    "rounded_intersection_3d": __import__("codecad_amd").shapes.intersection(
        [box(3).rotated((1, 1, 0), 30), sphere(3.6)], r=0.3),
    "rounded_union_2d": union([circle(2).translated_x(0.8), rectangle(2, 1)], r=0.25),
    "rounded_nested": union([union([box(1).translated_x(-1), box(1).translated_x(1)], r=0.2), sphere(1.2)], r=0.3)
    - cylinder(h=4, d=0.6),
}

"""Involute gears (reference shapes/gears.py:13-90)."""
import math

from .. import util
from . import base, simple2d


class InvoluteGearBase(base.Shape2D):
    """Unit-pitch-radius external involute profile with sharp tips and no root land."""

    def __init__(self, tooth_count, pressure_angle):
        self.tooth_count = tooth_count
        self.pressure_angle = math.radians(pressure_angle)

    def bounding_box(self):
        return util.BoundingBox(util.Vector(-1.5, -1.5), util.Vector(1.5, 1.5))

    def feature_size(self):
        return 0.5 * math.pi / self.tooth_count  # half a tooth thickness

    def get_node(self, point, cache):
        return cache.make_node("involute_gear", [self.tooth_count, self.pressure_angle], [point])


class InvoluteGear(simple2d.Union2D):
    """External gear, or (internal=True) the negative of an internal gear.

    = (involute profile scaled to the pitch radius, offset by -backlash, clipped by the
    tip circle) united with the root circle.  Attributes as in the reference:
    n, module, addendum_modules, dedendum_modules, pressure_angle, backlash, clearance,
    internal, pitch_diameter, root_diameter and outside_diameter / inside_diameter.
    """

    def __init__(self, n, module, addendum_modules=1, dedendum_modules=1, pressure_angle=20,
                 backlash=0, clearance=0, internal=False):
        self.n, self.module = n, module
        self.addendum_modules, self.dedendum_modules = addendum_modules, dedendum_modules
        self.pressure_angle, self.backlash = pressure_angle, backlash
        self.clearance, self.internal = clearance, internal
        self.pitch_diameter = n * module
        pitch_radius = self.pitch_diameter / 2

        if internal:
            inner = pitch_radius - addendum_modules * module
            outer = pitch_radius + dedendum_modules * module + clearance
            self.inside_diameter, self.root_diameter = inner * 2, outer * 2
            backlash = -backlash
        else:
            inner = pitch_radius - dedendum_modules * module - clearance
            outer = pitch_radius + addendum_modules * module
            self.outside_diameter, self.root_diameter = outer * 2, inner * 2

        profile = InvoluteGearBase(n, pressure_angle).scaled(pitch_radius)
        if backlash != 0:
            profile = profile.offset(-backlash)
        super().__init__([profile & simple2d.Circle(r=outer), simple2d.Circle(r=inner)])

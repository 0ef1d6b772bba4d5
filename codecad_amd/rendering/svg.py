"""SVG export of a 2D shape's boundary polygons (reference rendering/svg.py:4-36)."""
from . import polygon2d


def svg_document(obj, polygons=None):
    polygons = polygon2d.polygon(obj) if polygons is None else polygons
    box = obj.bounding_box()
    size = box.size()
    out = ['<svg xmlns="http://www.w3.org/2000/svg" ',
           'width="{}mm" height="{}mm" '.format(size.x, size.y),
           'viewBox="{} {} {} {}">'.format(box.a.x, -box.b.y, size.x, size.y),
           '<style type="text/css">path{stroke:#000;stroke-width:1px;vector-effect:non-scaling-stroke;fill:#BBF23C;}</style>',
           '<path d="']
    for polygon in polygons:
        it = reversed(polygon)   # y is flipped, so the winding is too
        x, y = next(it)
        out.append("M{},{}".format(x, -y))
        for x, y in it:
            out.append("L{},{}".format(x, -y))
        out.append("L{},{}".format(polygon[-1][0], -polygon[-1][1]))
    out.append('"/></svg>')
    return "".join(out)


def render_svg(obj, filename):
    with open(filename, "w") as fp:
        fp.write(svg_document(obj))

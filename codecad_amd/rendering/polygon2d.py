"""Boundary polygons of a 2D shape (reference rendering/polygon2d.py:36-173, kernel polygon2d.cl:82-175).

The reference walks the leaf blocks of a subdivision one at a time: grid_eval of the block's
corner samples, process_polygon over its triangular half cells, four blocking reads, then Python
follows the links.  Here the corner samples of ALL leaf blocks are evaluated in one launch
(grid_eval_blocks) and contoured in a second (hu_process_polygon_blocks); one read per array brings
the result back and `stitch` follows the links across blocks.
"""
import ctypes

import numpy

from .. import hip_util
from .. import subdivision
from ..hip_util import manager as hip_manager, check
from .. import grid_eval as _grid_eval

LINK_OVERFLOW_MASK = 0xfff00000   # polygon2d.py:11
_EMPTY = 0xffffffff


def _step_from_overflow_spec(spec):
    """Which neighbouring block a link leaves into (polygon2d.py:28-33)."""
    direction = -1 if spec & 0x20000000 else 1
    return (0, direction) if spec & 0x40000000 else (direction, 0)


def stitch(blocks, int_box_step):
    """Follow the links of every block into closed polygons.

    `blocks`: sequence of (int_corner (ix, iy), vertices (cells, 2), links (cells,), starts (n,)),
    one per leaf block.  Yields lists of (x, y) tuples.  A chain that leaves a block through its
    boundary continues in the neighbouring block at the start whose overflow spec matches
    (polygon2d.cl:161-163 flips the +/- bit so that both sides carry the same spec).  Polygons come
    out in block order, each starting at its lowest cell index (the reference's order follows its
    atomics and is unspecified); as cyclic sequences they are the reference's chains."""
    entry = {}
    for b, (corner, _vertices, _links, starts) in enumerate(blocks):
        for s in starts.tolist():
            key = (int(corner[0]), int(corner[1]), s & LINK_OVERFLOW_MASK)
            assert key not in entry, "two chains enter a block through the same boundary cell"
            entry[key] = (b, s & ~LINK_OVERFLOW_MASK)
    links = [numpy.array(blk[2], dtype=numpy.uint32, copy=True) for blk in blocks]
    for b0, (corner0, _v, _l, _s) in enumerate(blocks):
        live = numpy.flatnonzero(links[b0] != _EMPTY)
        for cell0 in live.tolist():
            if links[b0][cell0] == _EMPTY:
                continue
            chain = []
            b, cell = b0, cell0
            while True:
                nxt = int(links[b][cell])
                if nxt == _EMPTY:
                    assert (b, cell) == (b0, cell0), "a contour chain does not close"
                    break
                x, y = blocks[b][1][cell]
                chain.append((float(x), float(y)))
                links[b][cell] = _EMPTY   # visited
                if nxt & LINK_OVERFLOW_MASK:
                    spec = nxt & LINK_OVERFLOW_MASK
                    dx, dy = _step_from_overflow_spec(spec)
                    corner = blocks[b][0]
                    assert int_box_step is not None, "a contour leaves the only block"
                    key = (int(corner[0]) + dx * int_box_step, int(corner[1]) + dy * int_box_step, spec)
                    assert key in entry, "a contour leaves through a boundary no neighbouring block continues"
                    b, cell = entry[key]
                else:
                    cell = nxt
            yield chain


def contour_blocks(leaves, queue=None):
    """GPU part: -> (int_corners (n, 2) int, vertices (n, cells, 2) f32, links (n, cells) u32,
    starts list of u32 arrays), blocks sorted by integer corner."""
    queue = queue or hip_manager.queue
    gx, gy, gz = (int(d) for d in leaves.dims)
    assert gx < 512, "Larger grid size would overflow the index encoding"
    assert gz == 1
    n = leaves.count
    cells = (gx - 1) * (gy - 1) * 2
    per_block_starts = max((gx - 1) + (gy - 1), 1)
    if n == 0 or cells == 0:
        return numpy.zeros((0, 2), int), numpy.zeros((0, cells, 2), numpy.float32), numpy.zeros((0, cells), numpy.uint32), []
    corners = _grid_eval.grid_eval_blocks(leaves, pymcubes=False, queue=queue)
    vertices = hip_util.Buffer(numpy.float32, (n, cells, 2), queue=queue)
    links = hip_util.Buffer(numpy.uint32, (n, cells), queue=queue)
    starts = hip_util.Buffer(numpy.uint32, (n, per_block_starts), queue=queue)
    counters = hip_util.Buffer(numpy.uint32, (n,), queue=queue)
    counters.enqueue_fill(0)   # same in-order stream as the launches below
    d = (ctypes.c_uint32 * 2)(gx, gy)
    o = (ctypes.c_double * 3)(leaves.origin.x, leaves.origin.y, leaves.origin.z)
    ev = hip_util.Event(hip_manager, queue)
    check(hip_manager.lib.hu_process_polygon_blocks(corners.device_ptr, leaves.blocks.device_ptr, n,
                                                    float(leaves.resolution), o, numpy.float32(leaves.step), d,
                                                    vertices.device_ptr, links.device_ptr, starts.device_ptr,
                                                    counters.device_ptr, queue.handle), "hu_process_polygon_blocks")
    ev._done()
    host_blocks = numpy.empty((leaves.blocks.shape[0], 4), dtype=numpy.int32)
    leaves.blocks.read(out=host_blocks)
    v = vertices.read(wait_for=[ev]).copy()
    l = links.read().copy()
    s = starts.read().copy()
    c = counters.read().copy()
    for buf in (corners, vertices, links, starts, counters):
        buf.release()
    assert int(c.max()) <= per_block_starts
    ic = host_blocks[:n, :2].astype(int)
    order = numpy.lexsort((ic[:, 1], ic[:, 0]))
    return ic[order], v[order], l[order], [numpy.sort(s[i, :c[i]]) for i in order]


def polygon(obj, subdivision_grid_size=None):
    """Generate polygons (lists of (x, y)) representing the boundaries of a 2D shape."""
    obj.check_dimension(required=2)
    leaves = subdivision.subdivision_device(obj, obj.feature_size() / 2, grid_size=subdivision_grid_size)
    gx, gy, _ = (int(d) for d in leaves.dims)
    if leaves.count > 1:
        assert gx == gy
        int_box_step = leaves.int_step * (gx - 1)
    else:
        int_box_step = None   # no open chains if only one box is visited
    ic, vertices, links, starts = contour_blocks(leaves)
    leaves.blocks.release()
    yield from stitch([(ic[i], vertices[i], links[i], starts[i]) for i in range(len(ic))], int_box_step)


# ---- SVG export (reference rendering/svg.py:4-36) ------------------------------------------------
_SVG_STYLE = ('<style type="text/css">path{stroke:#000;stroke-width:1px;vector-effect:non-scaling-stroke;'
              'fill:#BBF23C;}</style>')


def svg_document(obj, polygons=None):
    """The SVG text of a 2D shape: one path, a sub-path per boundary polygon, in millimetres, y up."""
    polygons = polygon(obj) if polygons is None else polygons
    box = obj.bounding_box()
    width, height = box.size().x, box.size().y
    parts = ['<svg xmlns="http://www.w3.org/2000/svg" width="%smm" height="%smm" viewBox="%s %s %s %s">'
             % (width, height, box.a.x, -box.b.y, width, height), _SVG_STYLE, '<path d="']
    for chain in polygons:
        # flipping y flips the winding, so walk the chain backwards; close on its last point
        points = list(reversed(chain)) + [chain[-1]]
        parts.append("M%s,%s" % (points[0][0], -points[0][1]))
        parts.extend("L%s,%s" % (x, -y) for x, y in points[1:])
    parts.append('"/></svg>')
    return "".join(parts)


def render_svg(obj, filename):
    with open(filename, "w") as fp:
        fp.write(svg_document(obj))

"""Time the ray caster / bitmap kernels (1024x768, the reference's image test size) per shape and
count evaluate() passes, on the GPU; with --oracle also the CPU oracle.  Usage:
python tools/prof_render.py [--oracle] [names...]"""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shapes_zoo  # noqa: E402
from codecad_amd import nodes, hip_util  # noqa: E402
from codecad_amd.hip_util import manager as m  # noqa: E402
from codecad_amd.rendering import ray_caster, pictures  # noqa: E402
from codecad_amd import examples  # noqa: E402

SIZE = (1024, 768)


def time_shape(name, shape, reps=5):
    tape = nodes.make_program_buffer(shape)
    out = hip_util.Buffer(np.uint8, SIZE + (3,))
    if shape.dimension() == 2:
        origin, step = pictures.bitmap_arguments(shape, SIZE)
        launch = lambda: m.k.bitmap(SIZE, None, tape, origin.as_float4(), np.float32(step), out)
    else:
        cam = ray_caster.get_camera_params(shape.bounding_box(), SIZE, None)
        a = ray_caster.kernel_arguments(shape, *cam)
        launch = lambda: m.k.ray_caster(SIZE, None, tape, a["origin"].as_float4(), a["forward"].as_float4(),
                                        a["up"].as_float4(), a["right"].as_float4(), a["pixel_tolerance"],
                                        a["box_radius"], a["min_distance"], a["max_distance"], a["floor_z"], 0, out)
    launch().wait()
    best = 1e9
    for _ in range(reps):
        ev = launch()
        ev.wait()
        best = min(best, ev.elapsed_ms())
    out.release()
    return best, tape.instruction_count if hasattr(tape, "instruction_count") else 0


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    shapes = {**shapes_zoo.shapes_2d, **shapes_zoo.shapes_3d, "sponge4": examples.sponge(4), "csg_example": examples.csg_example()}
    for name in names or sorted(shapes):
        ms, _ = time_shape(name, shapes[name])
        line = "%-44s %8.3f ms  %7.1f Mpixel/s" % (name, ms, SIZE[0] * SIZE[1] / ms / 1e3)
        if "--oracle" in sys.argv:
            sys.path.insert(0, os.path.join(ROOT))
            import test_render_baselines as t
            if name in t.ALL:
                t0 = time.time()
                t.oracle_render(name, SIZE, threads=16)
                line += "   oracle(16 threads) %.0f ms" % ((time.time() - t0) * 1e3)
        print(line, flush=True)


if __name__ == "__main__":
    main()

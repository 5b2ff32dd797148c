"""The named workloads of BASELINE.json / SURVEY.md §8(d): scene + camera + harness light, as the
byte streams and launch parameters that the reference's main.cpp would produce for them."""
import numpy as np

from . import assets, hostlib

SCENES = {
    # name: (obj file, camera, add harness light)
    "room": ("room.obj", "reference", False),        # the only scene family the reference's own loader reads
    "teapot": ("teapot.obj", "auto", True),          # config #1
    "suzanne": ("suzanne.obj", "auto", True),        # config #2
    "scene": ("scene.obj", "auto", True),            # configs #3, #4 (the metric's scene)
    "testTable": ("testTable.obj", "auto", True),
    "triangle": ("triangle.obj", "auto", True),
}


class Workload:
    def __init__(self, name, width, height, with_qtable=False):
        obj, cam, light = SCENES[name]
        self.name, self.width, self.height = name, width, height
        self.scene = hostlib.Scene(assets.path(obj))
        pre = self.scene.info()
        # camera framing is fixed by the model itself, before the harness light is appended
        self.camera = dict(hostlib.REFERENCE_CAMERA) if cam == "reference" else hostlib.auto_camera(pre)
        if light and pre.num_emitters == 0:
            self.scene.add_harness_light()
        self.scene.build(with_qtable=with_qtable)
        self.info = self.scene.info()
        self.buffers = {s: self.scene.buffer(s) for s in hostlib.ALL_SLOTS}
        self.rays = hostlib.camera_rays(width=width, height=height, **self.camera)
        self.eye = np.array(self.camera["eye"], np.float32)

    @property
    def num_emitters(self):
        return self.info.num_emitters


class LatticeWorkload:
    """suzanne replicated on an nx x ny x nz jittered lattice (37 x 37 x 38 = 52 022
    copies = 50.36 M triangles at full size), random rotations, PCG32 seed 0x5EED, Kd 0.7, one emissive quad, flattened
    into one triangle stream.  `buffers` are views of the host library's memory (no copies), so the object must
    stay alive while they are in use."""

    def __init__(self, width, height, nx=37, ny=37, nz=38, seed=0x5EED, base="suzanne.obj"):
        self.name, self.width, self.height = f"lattice{nx}x{ny}x{nz}", width, height
        base_scene = hostlib.Scene(assets.path(base))
        base_scene.build(with_qtable=False)
        tris = base_scene.triangles().copy()
        base_scene.close()
        self.scene = hostlib.Scene()
        tris["mtlIndex"] = self.scene.add_material((0.7, 0.7, 0.7), 32.0, (0.0, 0.0, 0.0))
        tris["objId"] = 0
        self.scene.add_triangles(tris)
        self.scene.replicate_lattice(nx, ny, nz, seed)
        pre = self.scene.info()
        self.camera = hostlib.auto_camera(pre)
        self.scene.add_harness_light()
        self.scene.build(with_qtable=False)
        self.info = self.scene.info()
        self.buffers = {s: self.scene.buffer_view(s) for s in (3, 4, 5, 6)}
        self.rays = hostlib.camera_rays(width=width, height=height, **self.camera)
        self.eye = np.array(self.camera["eye"], np.float32)

    @property
    def num_emitters(self):
        return self.info.num_emitters


class TwoLightsWorkload:
    """A floor under two emissive quads of different size, height and power (4 emitter triangles: what the
    weights array of the reference's commented-out emitter pick, random.glsl:224, can hold).  Every surface point
    lies in front of at least one emitter, so that pick is defined everywhere (tests/golden/trace_twolights_*)."""

    def __init__(self, width, height):
        self.name, self.width, self.height = "twolights", width, height
        self.scene = hostlib.Scene()
        floor = np.zeros(2, hostlib.TRI_DTYPE)
        a, b, c, d = (-2.0, 0.0, -2.0), (-2.0, 0.0, 2.0), (2.0, 0.0, 2.0), (2.0, 0.0, -2.0)
        for k, (v0, v1, v2) in enumerate(((a, b, c), (a, c, d))):
            floor[k]["v0"], floor[k]["v1"], floor[k]["v2"] = v0, v1, v2
            floor[k]["n0"] = floor[k]["n1"] = floor[k]["n2"] = (0.0, 1.0, 0.0)
        floor["mtlIndex"] = self.scene.add_material((0.75, 0.6, 0.45), 24.0, (0.0, 0.0, 0.0))
        floor["objId"] = 0
        self.scene.add_triangles(floor)
        self.scene.add_quad_light((-0.8, 1.5, 0.0), 0.3, 0.3, (4.0, 3.5, 3.0))
        self.scene.add_quad_light((1.0, 0.8, 0.3), 0.15, 0.2, (0.553949, 0.914877, 1.0))
        self.scene.build(with_qtable=False)
        self.info = self.scene.info()
        self.camera = dict(eye=(0.0, 0.6, 2.5), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=60.0)
        self.buffers = {s: self.scene.buffer(s) for s in (3, 4, 5, 6)}
        self.rays = hostlib.camera_rays(width=width, height=height, **self.camera)
        self.eye = np.array(self.camera["eye"], np.float32)

    @property
    def num_emitters(self):
        return self.info.num_emitters


# ---------------------------------------------------------------------------------------------
# The analytic-scene kernels (reference trace.cs / trace1.cs / trace2.cs) have no scene buffers; their
# inputs are the camera and, for trace1 / trace2, the reference's sample / look-up tables.
ANALYTIC_CAMERAS = {
    "reference": dict(hostlib.REFERENCE_CAMERA),     # reference src/main.cpp:42-51: looks up, half the frame is sky
    "table": dict(eye=(2.5, 3.0, 4.0), look_at=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0), fov=60.0),  # sees table + pillar
    "light": dict(eye=(0.5, 1.5, -2.0), look_at=(0.0, 0.3, 3.0), up=(0.0, 1.0, 0.0), fov=60.0),  # faces trace1's light
}
TABLE_FILES = dict(sobol="sobol_256_256_4spp.data", scramble="scramble_128_128_8_4spp.data",
                   ranking="ranking_128_128_8_4spp.data", ltc1="ltc1.data", ltc2="ltc2.data")


def analytic_tables():
    """The table files Tracer::genBlueNoiseBuffers / genLtcData load (reference src/Tracer.cpp:23-90)."""
    return {k: assets.read(f) for k, f in TABLE_FILES.items()}


def analytic_camera(name, width, height):
    cam = ANALYTIC_CAMERAS[name]
    return hostlib.camera_rays(width=width, height=height, **cam), np.array(cam["eye"], np.float32)

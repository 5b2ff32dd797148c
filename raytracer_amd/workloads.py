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

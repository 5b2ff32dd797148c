"""Input data of the BASELINE.json configurations: scenes (OBJ/MTL), blue-noise sample tables and
LTC look-up tables.  They are data files of the reference's `res/` directory, kept gzip'd under
tests/golden/inputs/ because the reference tree is not present on the GPU box.  `path(name)`
unpacks one into a per-user cache directory and returns its path; scenes that come with a
material library are unpacked next to it."""
import gzip
import os
import tempfile

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUT_DIR = os.path.join(_ROOT, "tests", "golden", "inputs")
_COMPANIONS = {"room.obj": ["room.mtl"], "testTable.obj": ["testTable.mtl"]}


def _cache_dir():
    d = os.path.join(tempfile.gettempdir(), f"raytracer_amd_assets_{os.getuid()}")
    os.makedirs(d, exist_ok=True)
    return d


def _unpack(name):
    src = os.path.join(INPUT_DIR, name + ".gz")
    if not os.path.exists(src):
        raise FileNotFoundError(f"no input fixture {src}")
    dst = os.path.join(_cache_dir(), name)
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        with gzip.open(src, "rb") as f:
            data = f.read()
        tmp = dst + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, dst)
    return dst


def path(name):
    for extra in _COMPANIONS.get(name, []):
        _unpack(extra)
    return _unpack(name)


def read(name):
    with gzip.open(os.path.join(INPUT_DIR, name + ".gz"), "rb") as f:
        return f.read()


def available():
    return sorted(n[:-3] for n in os.listdir(INPUT_DIR) if n.endswith(".gz"))

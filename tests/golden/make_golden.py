#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference and Mesa llvmpipe; `make ref` first):
  * trace_*.npz / ids_*.npz / filter_*.npz : outputs of the reference's trace3.cs (+geometry, random,
    qtablehelper) and atrous.fs executed on llvmpipe by oracle/_ref/run_ref_shader, on scene byte
    streams built by this repository's host library from the input fixtures (tests/golden/inputs/);
  * kat_functions.npz : the reference's geometry.glsl / random.glsl functions called from
    oracle/probes/kat_reference_functions.comp;
  * semantics.npz     : oracle/probes/semantics.comp (how the oracle's GLSL evaluates built-ins);
  * camera_rays.json  : Tracer::updateRays arithmetic with the reference's vendored glm
    (oracle/_ref/ref_glm_camera).
The fixtures are data (inputs + expected outputs); no reference source text is stored.

  * analytic_*.npz    : the reference's trace.cs / trace1.cs / trace2.cs on llvmpipe (trace1 / trace2 with the
    load-time portability patches listed in oracle/run_ref_shader.c).
  * rl_sequence_*.npz : trace3.cs with main() calling its traceRL entry point, one path per dispatch.
  * lights_*.npz      : trace3.cs with random.glsl's emitter pick, live (uniform) and with its commented-out
    power / distance weighted block enabled at load time; radiance and the pick itself.
usage: python tests/golden/make_golden.py [--only trace|ids|filter|kat|semantics|camera|analytic|rl|lights]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refrun  # noqa: E402
from raytracer_amd import hostlib, workloads  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = refrun.REF_DIR
PROBE = os.path.join(ROOT, "oracle", "_ref", "glsl_probe")

# (scene, width, height, spp, max_bounces)
TRACE_CASES = [
    ("room", 96, 64, 2, 5),
    ("room", 96, 64, 1, 1),
    ("suzanne", 96, 54, 4, 5),
    ("scene", 96, 54, 4, 5),
    ("teapot", 80, 60, 1, 5),
    ("testTable", 64, 48, 2, 5),
    ("triangle", 48, 32, 1, 2),
]
# full-size primary-hit ID / visit-count frames of the BASELINE.json configurations
ID_CASES = [("teapot", 640, 480), ("suzanne", 1920, 1080), ("scene", 1920, 1080), ("room", 1920, 1080)]
# (kernel, camera, width, height, spp, uniforms) for the analytic-scene kernels trace.cs / trace1.cs / trace2.cs
ANALYTIC_CASES = [
    ("trace", "reference", 160, 120, 2, {}),
    ("trace", "table", 160, 120, 3, dict(importance=0)),
    ("trace", "table", 128, 96, 2, dict(phong_exponent=32.0, specular_factor=0.5)),
    ("trace1", "reference", 160, 120, 2, {}),
    ("trace1", "table", 160, 120, 3, {}),
    ("trace1", "table", 128, 96, 2, dict(mis=0)),
    ("trace1", "table", 128, 96, 2, dict(use_blue_noise=0)),
    ("trace1", "table", 256, 256, 1, {}),   # spans 128-pixel blue-noise tiles, incl. their out-of-range table reads
    ("trace1", "light", 128, 96, 2, {}),
    ("trace2", "reference", 160, 120, 1, {}),
    ("trace2", "table", 160, 120, 1, {}),
    ("trace2", "table", 128, 96, 1, dict(roughness=0.13)),
]
FILTER_CASES = [("room", 96, 64, 2, 5, 5), ("scene", 96, 54, 4, 5, 5), ("suzanne", 96, 54, 1, 1, 2)]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def buffers_digest(wl):
    return {str(s): sha(wl.buffers[s]) for s in (3, 4, 5, 6)}


def run_trace(wl, spp, mb, **kw):
    return refrun.run(wl.buffers, wl.width, wl.height, wl.rays, wl.eye, spp=spp, max_bounces=mb,
                      num_emitters=wl.num_emitters, **kw)


def gen_trace():
    for name, w, h, spp, mb in TRACE_CASES:
        wl = workloads.Workload(name, w, h)
        r = run_trace(wl, spp, mb)
        meta = dict(scene=name, width=w, height=h, spp=spp, max_bounces=mb, time0=1.25, dtime=0.0625,
                    num_emitters=wl.num_emitters, buffers_sha256=buffers_digest(wl))
        path = os.path.join(OUT, f"trace_{name}_{w}x{h}_{spp}spp_mb{mb}.npz")
        np.savez_compressed(path, color=r["color"], normal=r["normal"], pos=r["pos"].view(np.uint16),
                            tridx=r["tridx"], visits=r["visits"], stats=r["stats"], rays=wl.rays, eye=wl.eye,
                            meta=json.dumps(meta))
        print("wrote", path, os.path.getsize(path), "hit px", int((r["tridx"] >= 0).sum()))


def gen_ids():
    for name, w, h in ID_CASES:
        wl = workloads.Workload(name, w, h)
        r = run_trace(wl, 1, 0)  # primary hit only: independent of spp and bounce count
        tr, vi = r["tridx"], r["visits"]
        meta = dict(scene=name, width=w, height=h, tridx_sha256=sha(tr.tobytes()), visits_sha256=sha(vi.tobytes()),
                    normal_sha256=sha(r["normal"].tobytes()), pos_sha256=sha(r["pos"].tobytes()),
                    hit_pixels=int((tr >= 0).sum()), visits_sum=int(vi.sum()), tests_sum=int(r["stats"][..., 2].sum()),
                    buffers_sha256=buffers_digest(wl), trace_ms=r["trace_ms"])
        path = os.path.join(OUT, f"ids_{name}_{w}x{h}.npz")
        assert tr.min() >= -1 and tr.max() < 32768 and vi.max() < 32768
        np.savez_compressed(path, tridx=tr.astype(np.int16), visits=vi.astype(np.int16), rays=wl.rays, eye=wl.eye,
                            meta=json.dumps(meta))
        print("wrote", path, os.path.getsize(path), meta["hit_pixels"], meta["trace_ms"])


def analytic_name(kernel, cam, w, h, spp, uni):
    tag = "".join(f"_{k}{v:g}" for k, v in sorted(uni.items()))
    return f"analytic_{kernel}_{cam}_{w}x{h}_{spp}spp{tag}.npz"


def gen_analytic():
    tables = workloads.analytic_tables()
    for kernel, cam, w, h, spp, uni in ANALYTIC_CASES:
        rays, eye = workloads.analytic_camera(cam, w, h)
        r = refrun.run({}, w, h, rays, eye, spp=spp, shader=kernel, tables=tables, **uni)
        meta = dict(kernel=kernel, camera=cam, width=w, height=h, spp=spp, time0=1.25, dtime=0.0625, uniforms=uni)
        path = os.path.join(OUT, analytic_name(kernel, cam, w, h, spp, uni))
        np.savez_compressed(path, color=r["color"], normal=r["normal"], pos=r["pos"].view(np.uint16),
                            tridx=r["tridx"].astype(np.int16), rays=rays, eye=eye, meta=json.dumps(meta))
        print("wrote", path, os.path.getsize(path), "ids", np.unique(r["tridx"]).tolist())


def rl_sequence(wl, n, seed):
    """n single-path dispatches: rays through random points of the reference camera's image plane."""
    rng = np.random.default_rng(seed)
    rays = hostlib.camera_rays(width=64, height=64, **wl.camera)
    uv = rng.random((n, 2)).astype(np.float32)
    one = np.float32(1)
    d = ((rays[0] * (one - uv[:, 1:2]) + rays[1] * uv[:, 1:2]) * (one - uv[:, 0:1]) +
         (rays[2] * (one - uv[:, 1:2]) + rays[3] * uv[:, 1:2]) * uv[:, 0:1]).astype(np.float32)
    seq = np.zeros((n, 8), np.float32)
    seq[:, 0:3] = wl.eye
    seq[:, 3:6] = d
    seq[:, 6] = np.float32(1.25) + np.float32(0.0625) * np.arange(n, dtype=np.float32)
    return seq


def gen_rl():
    """traceRL (trace3.cs:695-791, dead in the reference's main): trace3.cs with main() calling traceRL, run one path
    per dispatch on a 1x1 image so that the order of the (non-atomic) Q-table updates is defined."""
    for name, n, mb, seed in (("room", 600, 5, 3), ("room", 200, 2, 4)):
        wl = workloads.Workload(name, 1, 1, with_qtable=True)
        seq = rl_sequence(wl, n, seed)
        r = refrun.run(wl.buffers, 1, 1, np.tile(seq[0, 3:6], (4, 1)), wl.eye, spp=1, max_bounces=mb,
                       num_emitters=wl.num_emitters, shader="trace3rl", seq=seq, instrument=False)
        q0 = np.frombuffer(wl.buffers[9], np.float32)
        changed = np.nonzero(r["qtable"].view(np.uint32) != q0.view(np.uint32))[0].astype(np.int32)
        visited = np.nonzero(r["qvisits"])[0].astype(np.int32)
        meta = dict(scene=name, n=n, max_bounces=mb, seed=seed, num_emitters=wl.num_emitters,
                    buffers_sha256={str(s): sha(wl.buffers[s]) for s in (3, 4, 5, 6, 7, 8, 9, 10)})
        path = os.path.join(OUT, f"rl_sequence_{name}_{n}_mb{mb}.npz")
        np.savez_compressed(path, seq=seq, seqcolor=r["seqcolor"].reshape(n, 4), q_changed_idx=changed,
                            q_changed_val=r["qtable"][changed], visits_idx=visited, visits_val=r["qvisits"][visited],
                            meta=json.dumps(meta))
        print("wrote", path, os.path.getsize(path), "updates", int(r["qvisits"].sum()), "cells", visited.size)


def gen_lights():
    """The emitter pick of randomEmitterSample (random.glsl:221-251), uniform (live) and weighted (its commented-out
    block, enabled at load time by run_ref_shader's weighted_lights=1): radiance frames, and the pick itself
    (light_prob, emitter index, random number) of every pixel's first bounce through the light_pick_probe patch."""
    w, h, spp, mb = 64, 48, 2, 3
    wl = workloads.TwoLightsWorkload(w, h)
    kw = dict(spp=spp, max_bounces=mb, num_emitters=wl.num_emitters)
    arrays = {}
    for tag, weighted in (("uniform", False), ("weighted", True)):
        r = refrun.run(wl.buffers, w, h, wl.rays, wl.eye, weighted_lights=weighted, **kw)
        arrays[f"color_{tag}"] = r["color"]
        pr = refrun.run(wl.buffers, w, h, wl.rays, wl.eye, weighted_lights=weighted, light_pick_probe=True,
                        spp=1, max_bounces=mb, num_emitters=wl.num_emitters)
        arrays[f"pick_{tag}"] = pr["color"]
        if not weighted:
            arrays.update(tridx=r["tridx"], visits=r["visits"], normal=r["normal"], pos=r["pos"].view(np.uint16))
    meta = dict(scene="twolights", width=w, height=h, spp=spp, max_bounces=mb, time0=1.25, dtime=0.0625,
                num_emitters=wl.num_emitters, buffers_sha256=buffers_digest(wl))
    path = os.path.join(OUT, f"lights_twolights_{w}x{h}_{spp}spp_mb{mb}.npz")
    np.savez_compressed(path, rays=wl.rays, eye=wl.eye, meta=json.dumps(meta), **arrays)
    differ = int((arrays["pick_uniform"].view(np.uint32) != arrays["pick_weighted"].view(np.uint32)).any(-1).sum())
    print("wrote", path, os.path.getsize(path), "pixels whose pick differs between the two rules:", differ)


def gen_filter():
    for name, w, h, spp, mb, iters in FILTER_CASES:
        wl = workloads.Workload(name, w, h)
        r = run_trace(wl, spp, mb, filter_iters=iters)
        path = os.path.join(OUT, f"filter_{name}_{w}x{h}_{iters}it.npz")
        np.savez_compressed(path, color=r["color"], normal=r["normal"], pos=r["pos"].view(np.uint16),
                            passes=np.stack([p.view(np.uint16) for p in r["filter"]]),
                            meta=json.dumps(dict(scene=name, iterations=iters, c_phi0=3.3, n_phi0=1e-2, p_phi0=5.5)))
        print("wrote", path, os.path.getsize(path))
    # a synthetic case with NaN colours, image-edge taps and flat regions
    rng = np.random.default_rng(5)
    w, h = 40, 24
    color = rng.random((h, w, 4), dtype=np.float32) * 2
    color[..., 3] = 1
    color[3, 5, 0] = np.nan
    color[10:14, 20:30] = 0.25
    normal = np.zeros((h, w, 4), np.int8)
    normal[..., 2] = 127
    normal[:, w // 2:, 0] = 127
    normal[:, w // 2:, 2] = 0
    pos = (rng.random((h, w, 4), dtype=np.float32) * 4).astype(np.float16)
    pos[..., 3] = 1
    wl = workloads.Workload("triangle", w, h)
    r = refrun.run(wl.buffers, w, h, wl.rays, wl.eye, spp=0, filter_iters=5, init_color=color, init_normal=normal,
                   init_pos=pos, instrument=False)
    path = os.path.join(OUT, "filter_synthetic_40x24_5it.npz")
    np.savez_compressed(path, color=color, normal=normal, pos=pos.view(np.uint16),
                        passes=np.stack([p.view(np.uint16) for p in r["filter"]]),
                        meta=json.dumps(dict(scene="synthetic", iterations=5, c_phi0=3.3, n_phi0=1e-2, p_phi0=5.5)))
    print("wrote", path, os.path.getsize(path))


def kat_inputs(n, seed=11):
    """Inputs that exercise hits, misses and edge cases of the box / triangle tests and samplers."""
    rng = np.random.default_rng(seed)
    f = np.float32
    inp = np.zeros((n, 32), f)
    origin = (rng.normal(size=(n, 3)) * 2).astype(f)
    v0 = (rng.normal(size=(n, 3)) * 2).astype(f)
    v1 = (v0 + rng.normal(size=(n, 3))).astype(f)
    v2 = (v0 + rng.normal(size=(n, 3))).astype(f)
    # aim most rays at a random point of (a slightly enlarged) triangle so that many hit, some graze
    a, b = rng.random(n) * 1.2 - 0.1, rng.random(n) * 1.2 - 0.1
    target = v0 + a[:, None] * (v1 - v0) + b[:, None] * (v2 - v0)
    d = target - origin
    miss = rng.random(n) < 0.2
    d[miss] = rng.normal(size=(int(miss.sum()), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(f)
    k = n // 16
    d[:k, 0] = 0.0            # axis-parallel rays: 1/0 = inf in the slab test
    d[k:2 * k, 1] = -0.0
    # rays through triangle vertices / along edges (exact-tie candidates)
    d[2 * k:3 * k] = (v0[2 * k:3 * k] - origin[2 * k:3 * k])
    d[3 * k:4 * k] = ((v0[3 * k:4 * k] + v1[3 * k:4 * k]) * f(0.5) - origin[3 * k:4 * k])
    bmin = np.minimum(np.minimum(v0, v1), v2) - (rng.random((n, 3)) * 0.2).astype(f)
    bmax = np.maximum(np.maximum(v0, v1), v2) + (rng.random((n, 3)) * 0.2).astype(f)
    origin[4 * k:5 * k, 0] = bmin[4 * k:5 * k, 0]   # origin on a slab plane
    origin[5 * k:6 * k] = (bmin[5 * k:6 * k] + bmax[5 * k:6 * k]) * f(0.5)  # origin inside the box
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[:8] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0], [0.6, 0, 0.8], [0.8, 0, 0.6]]
    inp[:, 0:3], inp[:, 3:6], inp[:, 6:9], inp[:, 9:12], inp[:, 12:15] = origin, d, v0, v1, v2
    inp[:, 15:18], inp[:, 18:21] = bmin, bmax
    inp[:, 21:24] = rng.random((n, 3), dtype=f)
    inp[:, 24:27] = nrm.astype(f)
    inp[:, 27] = np.choose(rng.integers(0, 6, n), [0.0, 18.062511, 35.999996, 225.0, 624.99997, 32.0]).astype(f)
    inp[:, 28:31] = (rng.normal(size=(n, 3)) * 3).astype(f)
    return inp


def run_probe(shaders, inp, words_per_item, groups):
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.f32"), os.path.join(tmp, "out.u32")
        np.ascontiguousarray(inp, np.float32).tofile(fin)
        n = inp.shape[0]
        subprocess.run([PROBE, ",".join(shaders), fin, str(n * words_per_item), fout, str(groups)], check=True)
        return np.fromfile(fout, np.uint32).reshape(n, words_per_item)


def gen_kat():
    n = 2048
    inp = kat_inputs(n)
    out = run_probe([os.path.join(ROOT, "oracle", "probes", "kat_reference_functions.comp"),
                     os.path.join(REF, "src", "shaders", "geometry.glsl"),
                     os.path.join(REF, "src", "shaders", "random.glsl")], inp, 48, n // 64)
    path = os.path.join(OUT, "kat_functions.npz")
    np.savez_compressed(path, inputs=inp, outputs=out[:, :28])
    print("wrote", path, os.path.getsize(path), "tri hits", int(out[:, 0].sum()), "box hits", int(out[:, 4].sum()))


def gen_semantics():
    rng = np.random.default_rng(7)
    inp = np.zeros((64, 16), np.float32)
    inp[:, :12] = rng.normal(size=(64, 12)).astype(np.float32) * 3
    inp[:, 12:] = rng.random((64, 4), dtype=np.float32)
    out = run_probe([os.path.join(ROOT, "oracle", "probes", "semantics.comp")], inp, 64, 1)
    path = os.path.join(OUT, "semantics.npz")
    np.savez_compressed(path, inputs=inp, outputs=out.view(np.float32))
    print("wrote", path)


def gen_camera():
    cams = [dict(eye=(0.0, 2.2, 3.5), look_at=(0.0, 3.0, 0.0), up=(0.0, 1.0, 0.0), fov=60.0, w=512, h=512),
            dict(eye=(0.0, 2.2, 3.5), look_at=(0.0, 3.0, 0.0), up=(0.0, 1.0, 0.0), fov=60.0, w=1920, h=1080),
            dict(eye=(1.5, -0.25, 7.0), look_at=(0.1, 0.6, 0.0), up=(0.0, 1.0, 0.0), fov=45.0, w=640, h=480),
            dict(eye=(-3.0, 4.0, 1.0), look_at=(0.5, 0.5, -0.5), up=(0.0, 1.0, 0.0), fov=75.0, w=3840, h=2160)]
    for name in ("teapot", "suzanne", "scene"):
        wl = workloads.Workload(name, 1920, 1080)
        cams.append(dict(eye=wl.camera["eye"], look_at=wl.camera["look_at"], up=wl.camera["up"], fov=60.0,
                         w=1920, h=1080))
    out = []
    for c in cams:
        args = [*c["eye"], *c["look_at"], *c["up"], c["fov"], c["w"], c["h"], 1.0, 2.0]
        txt = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_glm_camera")] + [repr(float(a)) for a in args],
                             capture_output=True, text=True, check=True).stdout
        rays = [[float.fromhex(t) for t in line.split()] for line in txt.strip().splitlines()]
        out.append(dict(camera=c, rays_hex=[[float(np.float32(v)).hex() for v in r] for r in rays]))
    with open(os.path.join(OUT, "camera_rays.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote camera_rays.json", len(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    if not refrun.available():
        sys.exit("reference oracle unavailable: run `make ref` in the container that holds /root/reference")
    steps = dict(camera=gen_camera, semantics=gen_semantics, kat=gen_kat, trace=gen_trace, filter=gen_filter,
                 ids=gen_ids, analytic=gen_analytic, rl=gen_rl, lights=gen_lights)
    for k, fn in steps.items():
        if a.only in (None, k):
            fn()
    log = refrun.run({}, 8, 8, np.zeros((4, 3), np.float32), (0, 0, 0), spp=0, instrument=False)["log"]
    manifest = dict(gl=[l for l in log.splitlines() if l.startswith("GL_")],
                    inputs={n: sha(open(os.path.join(OUT, "inputs", n), "rb").read())
                            for n in sorted(os.listdir(os.path.join(OUT, "inputs")))})
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()

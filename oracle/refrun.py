"""TEST INFRASTRUCTURE — drives oracle/_ref/run_ref_shader (the reference's GLSL on Mesa llvmpipe).

Only usable where /root/reference and swrast_dri.so exist (the build container); used by
tests/golden/make_golden.py to generate committed fixtures and by tests marked `ref`.
Never imported by the product path.
"""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RUNNER = os.path.join(HERE, "_ref", "run_ref_shader")
REF_DIR = os.environ.get("RT_REF_DIR", "/root/reference")


def available():
    return os.path.exists(RUNNER) and os.path.isdir(os.path.join(REF_DIR, "src", "shaders"))


def _hexf(x):
    return float(np.float32(x)).hex()


def run(scene_buffers, width, height, rays, eye, spp=1, frame0=0, time0=1.25, dtime=0.0625, max_bounces=5,
        num_emitters=0, importance=1, instrument=True, filter_iters=0, c_phi0=3.3, n_phi0=1e-2, p_phi0=5.5,
        repeat=1, init_color=None, init_normal=None, init_pos=None, stats_per_sample=False, timeout=3600,
        shader="trace3", phong_exponent=128.0, specular_factor=0.2, roughness=0.5, mis=1, use_blue_noise=1,
        tables=None, seq=None, weighted_lights=False, light_pick_probe=False):
    """scene_buffers: {slot: bytes} for slots 3..10 (missing → zeros). Returns dict of numpy arrays
    (row 0 = bottom image row, as GL) plus 'trace_ms' / 'filter_ms' lists and 'log'.
    shader: "trace3" (live kernel) | "trace" | "trace1" | "trace2"; tables: {"sobol"|"scramble"|"ranking"|"ltc1"|
    "ltc2": bytes} for trace1 / trace2 (the reference's res/*.data contents).
    shader "trace3rl" = trace3.cs with main() calling traceRL; seq = float32 [n,8] {eye, dir, time, blendFactor}:
    one dispatch per record with all corner rays = dir (use width = height = 1 for one path per dispatch)."""
    if not available():
        raise RuntimeError("reference oracle unavailable (no oracle/_ref/run_ref_shader or no reference dir)")
    names = {3: "nodes", 4: "tris", 5: "mtls", 6: "emitters", 7: "vnormals", 8: "vcentres", 9: "qtable",
             10: "qvisits"}
    with tempfile.TemporaryDirectory(prefix="rtref_") as tmp:
        lines = [f"ref_dir={REF_DIR}", f"width={width}", f"height={height}", f"spp={spp}", f"frame0={frame0}",
                 f"time0={_hexf(time0)}", f"dtime={_hexf(dtime)}", f"max_bounces={max_bounces}",
                 f"num_emitters={num_emitters}", f"importance={int(importance)}", f"instrument={int(instrument)}",
                 f"filter_iters={filter_iters}", f"c_phi0={_hexf(c_phi0)}", f"n_phi0={_hexf(n_phi0)}",
                 f"p_phi0={_hexf(p_phi0)}", f"repeat={repeat}", f"out={tmp}/o",
                 f"dump_stats_per_sample={int(stats_per_sample)}", f"shader={shader}",
                 f"phong_exponent={_hexf(phong_exponent)}", f"specular_factor={_hexf(specular_factor)}",
                 f"roughness={_hexf(roughness)}", f"mis={int(mis)}", f"use_blue_noise={int(use_blue_noise)}",
                 f"weighted_lights={int(weighted_lights)}", f"light_pick_probe={int(light_pick_probe)}"]
        for name, b in (tables or {}).items():
            p = os.path.join(tmp, name + ".tbl")
            with open(p, "wb") as f:
                f.write(bytes(b))
            lines.append(f"{name}={p}")
        for slot, name in names.items():
            b = scene_buffers.get(slot, b"") if scene_buffers else b""
            if b:
                p = os.path.join(tmp, name + ".bin")
                with open(p, "wb") as f:
                    f.write(b)
                lines.append(f"{name}={p}")
        rays = np.asarray(rays, np.float32).reshape(4, 3)
        for k, nm in enumerate(["ray00", "ray01", "ray10", "ray11"]):
            lines.append(f"{nm}=" + ",".join(_hexf(v) for v in rays[k]))
        lines.append("eye=" + ",".join(_hexf(v) for v in eye))
        for key, arr, dt in (("init_color", init_color, np.float32), ("init_normal", init_normal, np.int8),
                             ("init_pos", init_pos, np.float16)):
            if arr is not None:
                p = os.path.join(tmp, key + ".bin")
                np.ascontiguousarray(arr, dtype=dt).tofile(p)
                lines.append(f"{key}={p}")
        if seq is not None:
            sp = os.path.join(tmp, "seq.f32")
            np.ascontiguousarray(seq, np.float32).reshape(-1, 8).tofile(sp)
            lines.append(f"seq={sp}")
        job = os.path.join(tmp, "job.txt")
        with open(job, "w") as f:
            f.write("\n".join(lines) + "\n")
        env = dict(os.environ)
        env.setdefault("LIBGL_ALWAYS_SOFTWARE", "1")
        proc = subprocess.run([RUNNER, job], capture_output=True, text=True, timeout=timeout, env=env)
        if proc.returncode != 0:
            raise RuntimeError(f"run_ref_shader exit {proc.returncode}\n{proc.stdout}\n{proc.stderr}")
        res = {"log": proc.stdout + proc.stderr}
        res["trace_ms"] = [float(l.split()[-1]) for l in proc.stdout.splitlines() if l.startswith("trace rep")]
        res["filter_ms"] = [float(l.split()[-1]) for l in proc.stdout.splitlines() if l.startswith("filter rep")]

        def load(suffix, dt, ch):
            p = f"{tmp}/o{suffix}"
            if not os.path.exists(p):
                return None
            a = np.fromfile(p, dtype=dt)
            return a.reshape(height, width, ch) if ch > 1 else a.reshape(height, width)

        res["color"] = load(".color.f32", np.float32, 4)
        res["normal"] = load(".normal.i8", np.int8, 4)
        res["pos"] = load(".pos.f16", np.float16, 4)
        res["tridx"] = load(".tridx.i32", np.int32, 1)
        res["visits"] = load(".visits.i32", np.int32, 1)
        res["stats"] = load(".stats.i32", np.int32, 4)
        for suffix, key, dt in ((".qtable.f32", "qtable", np.float32), (".qvisits.i32", "qvisits", np.int32),
                                (".seqcolor.f32", "seqcolor", np.float32)):
            fp = f"{tmp}/o{suffix}"
            res[key] = np.fromfile(fp, dtype=dt) if os.path.exists(fp) else None
        res["filter"] = []
        for i in range(filter_iters):
            res["filter"].append(load(f".filter{i}.f16", np.float16, 4))
        if stats_per_sample:
            res["stats_per_sample"] = [load(f".stats{k}.i32", np.int32, 4) for k in range(spp)]
        return res

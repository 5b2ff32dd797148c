"""The drop-in class surface on a real MI355X: raytracer_amd/lib/rt_headless — the reference's main() (reference
src/main.cpp:32-134) built against the class mirror of raytracer_amd/host/ (Tracer, ComputeShader, FrameBuffer, Filter,
BVHContainer, ModelLoader, QTable, Camera) — run as a fresh child process, its raw image dumps checked against the
CPU oracle / the llvmpipe fixtures exactly as the ctypes-driven tests check the C-ABI.
Bar: snorm8 normals and fp16 positions bit-exact, radiance within tests/util.py's tolerance, filter within 1 fp16 step."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import assets, hiplib, hostlib, workloads
from tests.util import ROOT, assert_radiance_close, half_ulp_diff, load_npz

pytestmark = pytest.mark.gpu

HEADLESS = os.path.join(ROOT, "raytracer_amd", "lib", "rt_headless")


def run_headless(tmp_path, obj, *args, timeout=120):
    assert os.path.exists(HEADLESS), "raytracer_amd/lib/rt_headless is missing (make headless)"
    prefix = str(tmp_path / "dump")
    cmd = [HEADLESS, assets.path(obj), "--out", str(tmp_path / "frame"), "--dump-raw", prefix] + [str(a) for a in args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return prefix, r.stdout


def load_dump(prefix, w, h, filtered):
    d = dict(color=np.fromfile(prefix + "_color.f32", np.float32).reshape(h, w, 4),
             normal=np.fromfile(prefix + "_normal.i8", np.int8).reshape(h, w, 4),
             pos=np.fromfile(prefix + "_pos.f16", np.float16).reshape(h, w, 4))
    if filtered:
        d["filtered"] = np.fromfile(prefix + "_filtered.f16", np.float16).reshape(h, w, 4)
    return d


def oracle_frames(wl, frames, mb, time0=1.25):
    from oracle import orclib
    return orclib.trace(orclib.Scene(wl.buffers), wl.width, wl.height, wl.rays, wl.eye, spp=frames, max_bounces=mb,
                        time0=time0)


def check_against_oracle(got, ref, filter_iters, what):
    from oracle import orclib
    assert np.array_equal(got["normal"], ref["normal"]), what
    assert np.array_equal(got["pos"].view(np.uint16), np.asarray(ref["pos"]).view(np.uint16)), what
    assert_radiance_close(ref["color"], got["color"], what)
    if filter_iters:
        want = orclib.filter_passes(got["color"], got["normal"], got["pos"], filter_iters)[-1].view(np.uint16)
        d = half_ulp_diff(want, got["filtered"].view(np.uint16))
        assert d.max() <= 1 and (d == 0).mean() > 0.995, (what, int(d.max()))


def test_headless_room_trace_and_filter_against_oracle(tmp_path):
    """room.obj, the reference's camera, 128 x 72, 2 frames, 5 bounces, 2 filter passes: what smoke() checks through
    ctypes, here through Tracer::trace / Filter::filter / FrameBuffer of the C++ mirror."""
    W, H = 128, 72
    prefix, out = run_headless(tmp_path, "room.obj", "--size", f"{W}x{H}", "--frames", 2, "--bounces", 5, "--filter", 2)
    got = load_dump(prefix, W, H, True)
    wl = workloads.Workload("room", W, H)
    check_against_oracle(got, oracle_frames(wl, 2, 5), 2, "headless room")
    assert "Mrays/s" in out
    for name in ("frame.png", "frame_linear.png"):
        with open(tmp_path / name, "rb") as f:
            assert f.read(8) == b"\x89PNG\r\n\x1a\n"


def test_headless_tolerant_loader_auto_camera_and_light(tmp_path):
    """scene.obj (the metric's scene; the reference's own loader crashes on it) with --auto-camera --light = what
    workloads.Workload('scene') builds through the host C-ABI."""
    W, H = 160, 90
    prefix, _ = run_headless(tmp_path, "scene.obj", "--size", f"{W}x{H}", "--frames", 2, "--bounces", 3, "--auto-camera",
                             "--light")
    got = load_dump(prefix, W, H, False)
    wl = workloads.Workload("scene", W, H)
    check_against_oracle(got, oracle_frames(wl, 2, 3), 0, "headless scene")
    assert (got["normal"].astype(int) != 0).any()


@pytest.mark.parametrize("kernel,fixture", [("trace1", "analytic_trace1_reference_160x120_2spp.npz"),
                                            ("trace2", "analytic_trace2_reference_160x120_1spp.npz"),
                                            ("trace", "analytic_trace_reference_160x120_2spp.npz")])
def test_headless_analytic_kernels_against_llvmpipe_fixtures(tmp_path, kernel, fixture):
    g = load_npz(fixture)
    m = g["meta"]
    res = os.path.dirname(assets.path("ltc1.data"))
    for t in workloads.TABLE_FILES.values():
        assets.path(t)
    prefix, _ = run_headless(tmp_path, "room.obj", "--size", f"{m['width']}x{m['height']}", "--frames", m["spp"],
                             "--kernel", kernel, "--res", res)
    got = load_dump(prefix, m["width"], m["height"], False)
    assert np.array_equal(got["normal"], g["normal"])
    assert np.array_equal(got["pos"].view(np.uint16), np.asarray(g["pos"]).view(np.uint16))
    assert_radiance_close(g["color"], got["color"], fixture)


def test_headless_input_resets_accumulation(tmp_path):
    """every camera input resets frameNumber (reference src/glfw_setter.cpp:83-176), so the accumulation
    restarts.  6 frames with W pressed before frame 3 and COMMA (maxBounces + 1) before frame 4: the final image is
    frames 4 and 5 alone — camera one step forward, 4 bounces, blendFactor 0 then 1/2, time 1.25 + 0.0625 * {4, 5}."""
    W, H = 96, 64
    prefix, _ = run_headless(tmp_path, "room.obj", "--size", f"{W}x{H}", "--frames", 6, "--bounces", 3,
                             "--keys", "3:W,4:COMMA")
    got = load_dump(prefix, W, H, False)
    cam, fn, mb, it, fo, ex = hostlib.input_event(dict(hostlib.REFERENCE_CAMERA), 3, 3, 1, False, key=hostlib.KEYS["W"])
    assert fn == 0 and cam["eye"] != hostlib.REFERENCE_CAMERA["eye"]
    cam, fn, mb, it, fo, ex = hostlib.input_event(cam, 1, mb, it, fo, key=hostlib.KEYS["COMMA"])
    assert fn == 0 and mb == 4
    wl = workloads.Workload("room", W, H)
    wl.rays = hostlib.camera_rays(width=W, height=H, **cam)
    wl.eye = np.array(cam["eye"], np.float32)
    ref = oracle_frames(wl, 2, 4, time0=1.25 + 0.0625 * 4)
    check_against_oracle(got, ref, 0, "headless input script")
    # and it is NOT the un-reset accumulation from the original camera
    still = oracle_frames(workloads.Workload("room", W, H), 2, 3)
    assert not np.array_equal(got["pos"].view(np.uint16), np.asarray(still["pos"]).view(np.uint16))


def test_headless_freeze_time_and_reset(tmp_path):
    """--freeze-time: the time uniform keeps its first value, so every frame draws the same sample and the running
    average equals one frame; --reset-at K restarts the accumulation at frame K."""
    W, H = 64, 48
    wl = workloads.Workload("room", W, H)
    prefix, _ = run_headless(tmp_path, "room.obj", "--size", f"{W}x{H}", "--frames", 3, "--bounces", 2, "--freeze-time")
    got = load_dump(prefix, W, H, False)
    one = oracle_frames(wl, 1, 2)
    assert_radiance_close(one["color"], got["color"], "freeze-time")
    prefix, _ = run_headless(tmp_path, "room.obj", "--size", f"{W}x{H}", "--frames", 3, "--bounces", 2, "--reset-at", 2)
    got = load_dump(prefix, W, H, False)
    last = oracle_frames(wl, 1, 2, time0=1.25 + 0.0625 * 2)
    assert_radiance_close(last["color"], got["color"], "reset-at")


def test_headless_gpus_1_assembles_through_the_rccl_gather(tmp_path):
    """rt_headless --gpus 1: one GPU thread with its own context, an RCCL communicator of one rank (librccl opened at
    run time, ncclCommInitRank), tiles = the whole frame, rt_gather_tiles every frame.  The assembled frame equals the
    filtered frame of the plain run, bit for bit."""
    W, H = 128, 72
    args = ["--size", f"{W}x{H}", "--frames", 3, "--bounces", 3, "--filter", 3]
    prefix, _ = run_headless(tmp_path, "room.obj", *args)
    plain = load_dump(prefix, W, H, True)["filtered"]
    prefix, out = run_headless(tmp_path, "room.obj", *args, "--gpus", 1)
    assembled = np.fromfile(prefix + "_assembled.f16", np.float16).reshape(H, W, 4)
    assert np.array_equal(assembled.view(np.uint16), plain.view(np.uint16))
    assert "gpu 0: 128 x 72" in out


@pytest.mark.parametrize("args,ends_filtered", [(["--filter", 3, "--keys", "2:SPACE"], False),
                                                (["--filter", 0, "--keys", "1:SPACE,2:EQUAL"], True)])
def test_headless_gpus_1_follows_the_scripted_filter_state(tmp_path, args, ends_filtered):
    """SPACE toggles the filter and EQUAL / MINUS change its iterations WITHOUT restarting the accumulation (reference
    src/glfw_setter.cpp:207-218).  The multi-GPU path must gather the image the frame's scripted state names (not a
    process-wide static the GPU threads write at their own pace) and keep one tile grid / halo for the whole run: the
    assembled frame equals the single-context run's, bit for bit."""
    W, H = 128, 72
    base = ["--size", f"{W}x{H}", "--frames", 4, "--bounces", 3] + args
    prefix, _ = run_headless(tmp_path, "room.obj", *base)
    plain = load_dump(prefix, W, H, ends_filtered)
    prefix, _ = run_headless(tmp_path, "room.obj", *base, "--gpus", 1)
    if ends_filtered:
        assembled = np.fromfile(prefix + "_assembled.f16", np.float16).reshape(H, W, 4)
        assert np.array_equal(assembled.view(np.uint16), plain["filtered"].view(np.uint16))
    else:
        assembled = np.fromfile(prefix + "_assembled.f32", np.float32).reshape(H, W, 4)
        assert np.array_equal(assembled.view(np.uint32), plain["color"].view(np.uint32))


def test_headless_exits_nonzero_when_a_gpu_thread_fails(tmp_path):
    """--gpus 2 on a box with one GPU: the second thread cannot create its context; the first waits inside
    ncclCommInitRank for a peer that never comes.  rt_headless must report the failing rank and exit non-zero instead of
    hanging (it gives the other threads ten seconds, then leaves)."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one GPU")
    cmd = [HEADLESS, assets.path("room.obj"), "--out", str(tmp_path / "frame"), "--size", "64x48", "--frames", 2, "--gpus", 2]
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=90)
    assert r.returncode == 1, (r.returncode, r.stderr[-1500:])
    assert "NO USABLE GPU (rt_create(1)" in r.stderr and "gpu 1:" in r.stderr


def test_headless_blue_noise_flag(tmp_path):
    """rt_headless --blue-noise: Tracer::genBlueNoiseBuffers (reference src/Tracer.cpp:50-90) loads the three tables from
    --res, launchFlags carries RT_FLAG_BLUE_NOISE, frameIndex = frameNumber: the frames of the ctypes-driven oracle check."""
    W, H = 96, 64
    res = os.path.dirname(assets.path("ltc1.data"))
    for t in workloads.TABLE_FILES.values():
        assets.path(t)
    prefix, _ = run_headless(tmp_path, "room.obj", "--size", f"{W}x{H}", "--frames", 3, "--bounces", 4, "--blue-noise", "--res", res)
    got = load_dump(prefix, W, H, False)
    from oracle import orclib
    wl = workloads.Workload("room", W, H)
    tables = workloads.analytic_tables()
    ref = orclib.trace(orclib.Scene(wl.buffers), W, H, wl.rays, wl.eye, spp=3, max_bounces=4,
                       blue_noise={k: tables[k] for k in ("sobol", "scramble", "ranking")})
    check_against_oracle(got, ref, 0, "headless blue noise")


def test_headless_ltc_rectangle_lights_on_the_obj_scene(tmp_path):
    """rt_headless --kernel trace3ltc (RT_KERNEL_TRACE3_LTC, not a program of the reference): the C++ mirror's path — a
    Tracer built from "trace3ltc.cs", genLtcData from --res, Tracer::setRectLights, the `roughness` uniform — against the
    CPU oracle's same mode, with two filter passes on top."""
    from oracle import orclib
    from tests.test_gpu_ltc_bvh import room_lights
    W, H = 128, 72
    res = os.path.dirname(assets.path("ltc1.data"))
    assets.path("ltc2.data")
    lights = room_lights()
    args = []
    for l in lights.reshape(-1, 9):
        args += ["--rect-light", ",".join(repr(float(v)) for v in l)]
    prefix, _ = run_headless(tmp_path, "room.obj", "--size", f"{W}x{H}", "--frames", 2, "--filter", 2, "--kernel", "trace3ltc",
                             "--res", res, "--roughness", 0.25, "--light-intensity", 5.0, *args)
    got = load_dump(prefix, W, H, True)
    wl = workloads.Workload("room", W, H)
    ref = orclib.trace_ltc(orclib.Scene(wl.buffers), W, H, wl.rays, wl.eye, lights, workloads.analytic_tables(),
                           roughness=0.25, intensity=5.0)
    check_against_oracle(got, ref, 2, "headless trace3ltc")

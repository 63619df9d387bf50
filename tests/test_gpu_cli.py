"""GPU test of the `raytracer` CLI: the reference's flags and stdout contract
(src/main.cpp:9-28, src/render_wavefront.cpp:425-429), scraped with benchmark.py's own regular expressions
(benchmark.py:49-55), and out.png + the ray count compared BIT FOR BIT with the CPU oracle's render of the scene the C++
loader produced from the same file (loader -> scene description -> oracle; the GPU path is only on the CLI's side of the
comparison, so these are HIP-vs-oracle tests of rows f-1 / f-2, not HIP-vs-HIP)."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from rtamd import abi, scenes
from rtamd.glb_export import export_glb

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
EXE = REPO / "sycl-ray-tracer_amd" / "host" / "build" / "raytracer"


def _run(args, cwd):
    p = subprocess.run([str(EXE)] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = p.stdout
    # the exact patterns of the reference's benchmark.py
    rays_per_sec = float(re.search(r"Rays/sec: (\d+\.\d+)M", out).group(1))
    secs = float(re.search(r"Time measured: (\d+\.\d+) seconds", out).group(1))
    rays = int(re.search(r"Total rays: (\d+)", out).group(1))
    assert "Loading scene:" in out and "Writing image to disk" in out
    assert abs(rays / secs / 1e6 - rays_per_sec) <= 0.011 + 1e-3 * rays_per_sec
    return out, rays


def _oracle_frame(oracle, glb, w, h, kind, depth, spp):
    """What the reference's main() does with `glb`, on the CPU: load (C++ loader, the file's own camera node or the documented
    default pose), Camera ctor, render_frame. Returns (rgba_u8, rays, loaded scene)."""
    from rtamd import loader
    ld = loader.load_glb(glb)
    cam = oracle.camera(w, h, ld.camera.position, ld.camera.direction, ld.camera.focal_length)
    _, b, rays = oracle.OracleScene(ld).render(cam, kind, depth, spp, use_bvh=True)
    return b, rays, ld


@pytest.mark.parametrize("flag,kind", [("-m", abi.RT_RENDERER_MEGAKERNEL), ("-w", abi.RT_RENDERER_WAVEFRONT), (None, abi.RT_RENDERER_WAVEFRONT)])
def test_cli_renders_the_shipped_cube(rtlib, oracle, tmp_path, flag, kind):
    from PIL import Image
    w, h, depth, spp = 96, 64, 5, 3
    glb = REPO / "assets" / "cube.glb"
    args = ["-d", depth, "--sample-count", spp, "--width", w, "--height", h, "--quiet"] + ([flag] if flag else []) + [glb]
    out, rays = _run(args, tmp_path)  # default output name, in the working directory, as the reference does
    if kind == abi.RT_RENDERER_WAVEFRONT:
        assert [f"Sample {i}" in out for i in range(spp)] == [True] * spp  # src/render_wavefront.cpp:402
    exp_u8, exp_rays, _ = _oracle_frame(oracle, glb, w, h, kind, depth, spp)
    assert rays == exp_rays
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "out.png")), exp_u8)


@pytest.mark.parametrize("flag,kind", [("-m", abi.RT_RENDERER_MEGAKERNEL), ("-w", abi.RT_RENDERER_WAVEFRONT)])
def test_cli_on_an_exported_scene_with_camera_textures_and_extensions(rtlib, oracle, tmp_path, scene_cache, flag, kind):
    """Generated atrium -> GLB (instancing, rotations, camera node, emissive / ior extensions, nine PNG textures, sky extras)
    -> `raytracer`: PNG and ray count equal the oracle's render of the LOADED scene with the LOADED camera. (The camera
    node stores a quaternion, so the loaded direction differs from the generator's in the last bits; both sides of this
    comparison read it from the file, as the reference's main() does: src/main.cpp:48-55, src/scene.cpp:109-128.)"""
    from PIL import Image
    sd = scene_cache("atrium", detail=1)
    glb = tmp_path / "atrium1.glb"
    export_glb(sd, glb)
    w, h, depth, spp = 160, 90, 6, 2
    out, rays = _run([flag, "-d", depth, "-s", spp, "--width", w, "--height", h, "--out", tmp_path / "a.png", glb], tmp_path)
    assert "Loading 9 images" in out and "Baked 9 images into array" in out
    img = np.asarray(Image.open(tmp_path / "a.png"))
    assert img.shape == (h, w, 4) and (img[..., 3] == 255).all()
    exp_u8, exp_rays, ld = _oracle_frame(oracle, glb, w, h, kind, depth, spp)
    assert ld.n_triangles == sd.n_triangles and ld.textures.shape[0] == 9
    assert rays == exp_rays
    np.testing.assert_array_equal(img, exp_u8)


def test_cli_failures_are_reported_not_fatal(rtlib, tmp_path):
    p = subprocess.run([str(EXE), "-s", "1", str(tmp_path / "missing.glb")], capture_output=True, text=True)
    assert p.returncode == 1 and "Failed to load .glTF" in p.stdout


@pytest.mark.parametrize("flag", ["-m", "-w"])
def test_cli_tiles_a_frame_over_several_devices_in_one_process(rtlib, oracle, tmp_path, flag):
    """--devices A,B,C: interleaved 8-row strips, one host thread and one renderer per tile (here three tiles on the
    one GPU of the box, which exercises everything except a second physical device): the PNG and the ray count are
    those of the single-device run, and a device that does not exist is refused."""
    from PIL import Image
    common = ["-d", 6, "-s", 3, "--width", 200, "--height", 150, "--quiet", flag, REPO / "assets" / "cube.glb"]
    _, rays1 = _run(["--out", tmp_path / "one.png"] + common, tmp_path)
    out3, rays3 = _run(["--devices", "0,0,0", "--out", tmp_path / "three.png"] + common, tmp_path)
    assert "Running on devices: 0 0 0" in out3 and "Tile gather: device copies" in out3  # one GPU here: no RCCL rank per tile
    assert rays3 == rays1
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "three.png")), np.asarray(Image.open(tmp_path / "one.png")))
    exp_u8, exp_rays, _ = _oracle_frame(oracle, REPO / "assets" / "cube.glb", 200, 150,
                                        abi.RT_RENDERER_MEGAKERNEL if flag == "-m" else abi.RT_RENDERER_WAVEFRONT, 6, 3)
    assert rays3 == exp_rays
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "three.png")), exp_u8)  # gathered on the device, equal to the oracle
    p = subprocess.run([str(EXE), "--devices", "0,99", "-s", "1", str(REPO / "assets" / "cube.glb")], capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 1 and "does not exist" in p.stdout


@pytest.mark.parametrize("schedule,expect", [("per-bounce", "extend 45, shade 45, shoot 0, finish 0"), ("per-bounce-fused", "extend 0, shade 0, shoot 45, finish 0"),
                                             ("per-sample", "extend 0, shade 0, shoot 0, finish 6")])
def test_cli_schedule_flag_selects_and_reports_the_schedule(rtlib, oracle, tmp_path, schedule, expect):
    """--schedule: the wavefront renderer's other schedules through the CLI (rt_renderer_set_schedule in the C++ adapter); the line it
    prints comes from rt_stats (what ran: 3 samples x 5 bounces on 3 stream lanes; a launch per sample: 2 lanes), and PNG and ray count stay the oracle's."""
    from PIL import Image
    w, h, depth, spp = 96, 64, 5, 3
    glb = REPO / "assets" / "cube.glb"
    out, rays = _run(["-w", "-d", depth, "-s", spp, "--width", w, "--height", h, "--quiet", "--schedule", schedule, glb], tmp_path)
    assert f"launches: {expect}" in out, out
    exp_u8, exp_rays, _ = _oracle_frame(oracle, glb, w, h, abi.RT_RENDERER_WAVEFRONT, depth, spp)
    assert rays == exp_rays
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "out.png")), exp_u8)
    p = subprocess.run([str(EXE), "--schedule", "nonsense", str(glb)], capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 105

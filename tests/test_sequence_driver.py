"""The sequence driver examples/orb_tum (shape of Examples/Monocular/mono_tum.cc): list and settings
parsing on the CPU, and on the GPU a whole synthetic sequence of PGM/PPM files whose keypoints and
descriptors must equal the oracle's frame by frame."""
import os
import struct
import subprocess

import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, splitmix64, synth_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "orb_tum")

SETTINGS = """%YAML:1.0
# Camera Parameters
Camera.fx: 517.306408
Camera.RGB: 1
# ORB Parameters
ORBextractor.nFeatures: 500
ORBextractor.scaleFactor: 1.2
ORBextractor.nLevels: 6
ORBextractor.iniThFAST: 20
ORBextractor.minThFAST: 7
"""


def _gray_from_rgb(rgb, rgb_flag):
    """Tracking::GrabImageMonocular's conversion (src/Tracking.cc:189-197) of a file's R,G,B triplets: cv::imread
    stores B,G,R; RGB2GRAY takes channel 0 as red."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    c0, c1, c2 = b, g, r
    rr, bb = (c0, c2) if rgb_flag else (c2, c0)
    return ((rr * 4899 + c1 * 9617 + bb * 1868 + 8192) >> 14).astype(np.uint8)


def _write_sequence(d, n, w, h):
    os.makedirs(os.path.join(d, "rgb"), exist_ok=True)
    grays, lines = [], ["# color images", "# file: synthetic", "# timestamp filename"]
    for f in range(n):
        ts = 1305031102.175304 + f / 30.0
        if f % 3 == 2:  # a colour frame
            base = synth_scene(w, h, frame_seed(1, 40 + f)).astype(np.int64)
            noise = (splitmix64(900 + f, w * h * 3) % np.uint64(31)).astype(np.int64).reshape(h, w, 3) - 15
            rgb = np.clip(base[..., None] + noise, 0, 255).astype(np.uint8)
            name = "rgb/%.6f.ppm" % ts
            with open(os.path.join(d, name), "wb") as fh:
                fh.write(b"P6\n# synthetic\n%d %d\n255\n" % (w, h))
                fh.write(rgb.tobytes())
            grays.append(_gray_from_rgb(rgb, True))
        else:
            g = synth_scene(w, h, frame_seed(1, 40 + f))
            name = "rgb/%.6f.pgm" % ts
            with open(os.path.join(d, name), "wb") as fh:
                fh.write(b"P5 %d %d 255\n" % (w, h))
                fh.write(g.tobytes())
            grays.append(g)
        lines.append("%.6f %s" % (ts, name))
    with open(os.path.join(d, "rgb.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(d, "settings.yaml"), "w") as fh:
        fh.write(SETTINGS)
    return grays


def test_driver_parses_list_and_settings(tmp_path):
    if not os.path.exists(EXE):  # normally built by __graft_entry__.build()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "examples", "orb_tum.cc"),
                               "-L" + os.path.join(ROOT, "orb_slam2_amd", "csrc"), "-lorbhip",
                               "-Wl,-rpath,$ORIGIN/../orb_slam2_amd/csrc", "-o", EXE])
    d = str(tmp_path)
    _write_sequence(d, 4, 64, 64)
    out = subprocess.run([EXE, os.path.join(d, "settings.yaml"), d, os.path.join(d, "o.orbk"), "--list-only"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "- Number of Features: 500" in out.stdout and "- Scale Levels: 6" in out.stdout
    assert "- Initial Fast Threshold: 20" in out.stdout and "- Minimum Fast Threshold: 7" in out.stdout
    assert "Images in the sequence: 4" in out.stdout
    # a settings file without the extractor keys is refused, as cv::FileStorage would yield zeros
    bad = os.path.join(d, "bad.yaml")
    open(bad, "w").write("%YAML:1.0\nCamera.fx: 1.0\n")
    out = subprocess.run([EXE, bad, d, os.path.join(d, "o.orbk"), "--list-only"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0


@pytest.mark.gpu
def test_driver_sequence_equals_oracle(oracle, tmp_path):
    assert os.path.exists(EXE), "examples/orb_tum is built by __graft_entry__.build()"
    d = str(tmp_path)
    n, w, h = 7, 320, 240
    grays = _write_sequence(d, n, w, h)
    fout = os.path.join(d, "out.orbk")
    res = subprocess.run([EXE, os.path.join(d, "settings.yaml"), d, fout, "3"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(fout, "rb").read()
    magic, version, nframes = struct.unpack_from("<III", raw, 0)
    assert magic == 0x4B42524F and version == 1 and nframes == n
    off = 12
    e = oracle.Extractor(500, 1.2, 6, 20, 7, oracle.QUIRKS_FORK)
    for f in range(n):
        ts, fw, fh, cnt = struct.unpack_from("<dIII", raw, off)
        off += 20
        assert (fw, fh) == (w, h) and abs(ts - (1305031102.175304 + f / 30.0)) < 1e-5
        kps = np.frombuffer(raw, oracle.KP_DTYPE, cnt, off)
        off += cnt * 28
        desc = np.frombuffer(raw, np.uint8, cnt * 32, off).reshape(cnt, 32)
        off += cnt * 32
        ko, do = e.extract(grays[f])
        assert cnt == len(ko), "frame %d: %d keypoints, oracle %d" % (f, cnt, len(ko))
        assert (kps == ko).all() and (desc == do).all(), "frame %d differs from the oracle" % f
    assert off == len(raw)

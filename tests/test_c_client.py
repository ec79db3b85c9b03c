"""The drop-in boundary from plain C: examples/ekf_demo.c includes only include/ekf_monoslam.h and links only
libekfslam_hip.so (no torch, no Python, no C++ on the client side)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "ekf-monoslam_for_3d-reconstruction_amd", "lib")
SRC = os.path.join(ROOT, "examples", "ekf_demo.c")
SRC_CPP = os.path.join(ROOT, "examples", "vslam_filter_demo.cpp")


def _build(out):
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L", LIBDIR, "-lekfslam_hip", "-Wl,-rpath," + LIBDIR, "-lm"]
    return subprocess.run(cmd, capture_output=True, text=True)


def _build_cpp(out):
    cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC_CPP, "-o", out,
           "-L", LIBDIR, "-lekfslam_hip", "-Wl,-rpath," + LIBDIR]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_client_compiles_and_links_as_c99(tmp_path):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(LIBDIR, "libekfslam_hip.so")):
        g.build()
    r = _build(str(tmp_path / "ekf_demo"))
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = str(tmp_path / "ekf_demo")
    r = _build(exe)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.strip().endswith("ok")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_mirror_client_compiles(tmp_path):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(LIBDIR, "libekfslam_hip.so")):
        g.build()
    r = _build_cpp(str(tmp_path / "vslam_filter_demo"))
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_cpp_mirror_client_runs(tmp_path):
    """include/vslam_filter_hip.hpp driven like the node drives VSlamFilter: frame, templates, predict, matcher,
    1-point RANSAC, two-stage update, conversion, ellipses, map export."""
    exe = str(tmp_path / "vslam_filter_demo")
    r = _build_cpp(exe)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.strip().endswith("ok")

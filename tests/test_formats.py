"""On-disk formats of the reference node: Eigen default IOFormat restated."""
import io

import numpy as np

from __graft_entry__ import load_package

pkg = load_package()
from ekf_monoslam_amd import formats  # noqa: E402


def test_eigen_default_format_alignment_and_precision():
    # Eigen: width = widest coefficient, right-aligned, one space between columns, %g with 6 digits
    assert formats.format_eigen(np.array([[1.0, 2.5], [-3.0, 4.0]])) == "  1 2.5\n -3   4"
    assert formats.format_eigen(np.array([0.123456789, 1e-5, 123456789.0])) == "   0.123457\n      1e-05\n1.23457e+08"
    assert formats.format_eigen(np.array([[1, 20, 300]], np.int64)) == "  1  20 300"
    assert formats.format_eigen(np.zeros((0, 3))) == ""


def test_points_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    t = rng.normal(size=(17, 12)).astype(np.float32) * np.float32(3.0)
    p = tmp_path / "points.txt"
    formats.write_points(p, t)
    text = p.read_text()
    assert not text.endswith("\n") and len(text.splitlines()) == 17
    back = formats.read_points(p)
    assert back.shape == (17, 12)
    assert np.allclose(back, t, rtol=6e-6, atol=0)        # 6 significant digits


def test_pose_and_covariance_records():
    buf = io.StringIO()
    pose = np.array([0.1, -0.2, 0.3, 0.0, 0.0, -0.70710678, 0.70710678], np.float32)
    buf.write(formats.pose_record(12, pose, [[3, 101, 57], [9, 12, 200]]))
    buf.write(formats.pose_record(13, pose, None))
    text = buf.getvalue()
    lines = text.splitlines()
    assert lines[0] == "P12" and lines[8] == "  3 101  57" and lines[9] == "  9  12 200"
    assert lines[10] == "P13" and lines[18] == "0  0  0"
    recs = formats.read_pose_records(io.StringIO(text))
    assert [r[0] for r in recs] == [12, 13]
    assert np.allclose(recs[0][1], pose, rtol=6e-6) and recs[0][2].tolist() == [[3, 101, 57], [9, 12, 200]]
    assert recs[1][2].tolist() == [[0, 0, 0]]
    S = np.diag(np.arange(1, 15, dtype=np.float32))
    cov = formats.camera_cov_record(S) + formats.camera_cov_record(2 * S)
    blocks = formats.read_camera_covs(io.StringIO(cov))
    assert blocks.shape == (2, 7, 7) and np.allclose(blocks[1], 2 * S[:7, :7])

"""On-disk formats the reference's ROS node writes around the filter, restated
for the host side of the HIP path.  The node streams Eigen objects with the default
``Eigen::IOFormat`` (monoslam_ransac.cpp:232-236, 272-275, 638-658):

  points.txt            one N x 12 matrix: ``getPointsFeatures()`` (RosVSLAMRansac.cpp:340-418) =
                        ``ekf_export_points`` -- [x y z] * map_scale, the 3x3 covariance, row-major
  nodes_and_prjcts.txt  per kept pose: ``P<id>``, the 7 pose entries [r q] one per line, then the
                        projections (``Point4sba``, vR.cpp:1319-1336: real_index, u, v as ints) or
                        the literal ``0  0  0`` line
  cams_cov.txt          per kept pose: the 7x7 camera covariance block

Eigen's default format: coefficients written with the stream's precision (6 significant digits,
``%g`` style), right-aligned to the widest coefficient of the matrix, one space between columns,
newline between rows, nothing after the last row (the node adds ``endl``).
"""
from __future__ import annotations

import io
from typing import Iterable, Sequence

import numpy as np


def _coeff(x, precision: int) -> str:
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    return ("%." + str(precision) + "g") % float(x)


def format_eigen(mat, precision: int = 6) -> str:
    """``std::ostream << Eigen::Matrix`` with the default IOFormat (aligned columns)."""
    a = np.asarray(mat)
    if a.ndim == 1:
        a = a.reshape(-1, 1)                      # Eigen vectors are columns
    if a.size == 0:
        return ""
    cells = [[_coeff(v, precision) for v in row] for row in a.tolist()]
    width = max(len(c) for row in cells for c in row)
    return "\n".join(" ".join(c.rjust(width) for c in row) for row in cells)


def write_points(path_or_file, table) -> None:
    """points.txt: ``f_points << slam.getPointsFeatures()`` (monoslam_ransac.cpp:272-275), no trailing newline."""
    table = np.asarray(table)
    if table.ndim != 2 or table.shape[1] != 12:
        raise ValueError("points table must be N x 12")
    _write(path_or_file, format_eigen(table))


def read_points(path_or_file) -> np.ndarray:
    """Whitespace-separated reader (what sba_add.cpp:79-81 does); returns N x 12 float32."""
    text = _read(path_or_file)
    vals = np.array(text.split(), dtype=np.float32)
    if vals.size % 12:
        raise ValueError("points.txt does not hold rows of 12 numbers")
    return vals.reshape(-1, 12)


def pose_record(pose_id: int, state7: Sequence[float], projections=None) -> str:
    """One record of nodes_and_prjcts.txt (monoslam_ransac.cpp:638-652)."""
    s = np.asarray(state7, dtype=np.float32).reshape(-1)
    if s.size != 7:
        raise ValueError("pose = [r(3) q(4)]")
    if projections is None or len(projections) == 0:
        proj = "0  0  0"
    else:
        p = np.asarray(projections)
        if p.ndim != 2 or p.shape[1] != 3:
            raise ValueError("projections are rows of (real_index, u, v)")
        proj = format_eigen(p.astype(np.int64))    # MatrixX3i: z truncated to int (vR.cpp:1324)
    return "P%d\n%s\n%s\n" % (int(pose_id), format_eigen(s), proj)


def camera_cov_record(sigma) -> str:
    """One record of cams_cov.txt: ``cov_cams << Sigma.block<7,7>(0,0) << endl``."""
    a = np.asarray(sigma, dtype=np.float32)
    if a.shape[0] < 7 or a.shape[1] < 7:
        raise ValueError("needs at least the 7x7 pose block")
    return format_eigen(a[:7, :7]) + "\n"


def read_pose_records(path_or_file):
    """Parse nodes_and_prjcts.txt back: list of (id, pose7 float32, projections int64 (k x 3))."""
    lines = [ln for ln in _read(path_or_file).splitlines() if ln.strip()]
    out, i = [], 0
    while i < len(lines):
        if not lines[i].startswith("P"):
            raise ValueError("record must start with P<id>: %r" % lines[i])
        pid = int(lines[i][1:])
        pose = np.array([float(lines[i + 1 + k]) for k in range(7)], dtype=np.float32)
        i += 8
        rows = []
        while i < len(lines) and not lines[i].startswith("P"):
            rows.append([int(float(t)) for t in lines[i].split()])
            i += 1
        out.append((pid, pose, np.array(rows, dtype=np.int64).reshape(-1, 3)))
    return out


def read_camera_covs(path_or_file) -> np.ndarray:
    vals = np.array(_read(path_or_file).split(), dtype=np.float32)
    if vals.size % 49:
        raise ValueError("cams_cov.txt does not hold 7x7 blocks")
    return vals.reshape(-1, 7, 7)


def _write(path_or_file, text: str) -> None:
    if hasattr(path_or_file, "write"):
        path_or_file.write(text)
    else:
        with open(path_or_file, "w") as fh:
            fh.write(text)


def _read(path_or_file) -> str:
    if hasattr(path_or_file, "read"):
        return path_or_file.read()
    with open(path_or_file) as fh:
        return fh.read()

#!/usr/bin/env python3
"""Golden vectors of the image side from the CPU oracle `oracle/image_oracle.py`:
a 96 x 128 textured frame, 6 templates, blurred matching templates for given (h, hb) pairs and the
result of the NCC search in a moved frame.  Like config1_n20_*.npz they pin the ORACLE'S outputs
(the reference ships no fixtures): a regression guard on the CPU and an oracle-free check on the GPU box.

    python tests/golden/make_golden_image.py      # rewrites tests/golden/image_w15.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import image_oracle as io_  # noqa: E402

W = 15
CENTRES = [(30.3, 25.8), (64.0, 40.0), (100.9, 30.2), (40.5, 70.5), (85.2, 66.6), (112.4, 80.1)]
# (h, hb) pairs: no blur (short), horizontal, vertical, oblique both ways, long oblique
BLUR = [((30.0, 25.0), (30.5, 25.5)), ((64.0, 40.0), (58.0, 40.0)), ((100.0, 30.0), (100.0, 37.5)),
        ((40.5, 70.5), (46.2, 66.1)), ((85.2, 66.6), (79.9, 60.3)), ((112.4, 80.1), (101.3, 84.9))]
S_BLOCKS = [[[9.0, 1.0], [1.0, 6.0]], [[4.5, -0.8], [-0.8, 5.2]], [[30.0, 4.0], [4.0, 12.0]],
            [[6.0, 0.0], [0.0, 6.0]], [[0.3, 0.0], [0.0, 0.3]], [[150.0, 20.0], [20.0, 90.0]]]
H_PRED = [(31.2, 25.1), (63.1, 41.7), (99.2, 31.9), (41.9, 69.0), (86.0, 67.0), (110.0, 82.0)]


def main():
    frame = io_.random_texture(96, 128, seed=31)
    moved = np.roll(np.roll(frame, 2, axis=1), -1, axis=0)
    moved[55:80, 70:100] = 90                                  # hides template 4
    tpl = np.stack([io_.capture_patch(frame, u, v, W) for (u, v) in CENTRES])
    blurred = np.stack([io_.matching_patch(tpl[i], np.float32(BLUR[i][0]), np.float32(BLUR[i][1]), 2) for i in range(6)])
    found, z, score = [], [], []
    for i in range(6):
        ok, zz, sc, _ = io_.find_match(moved, tpl[i], np.float32(H_PRED[i]), np.array(S_BLOCKS[i], np.float32), 2)
        found.append(ok); z.append(zz); score.append(sc)
    out = dict(frame=frame, moved=moved, centres=np.array(CENTRES, np.float32), templates=tpl,
               blur_h=np.array([b[0] for b in BLUR], np.float32), blur_hb=np.array([b[1] for b in BLUR], np.float32),
               blurred=blurred, S=np.array(S_BLOCKS, np.float32), h_pred=np.array(H_PRED, np.float32),
               found=np.array(found), z=np.array(z, np.int32), score=np.array(score, np.float32))
    path = os.path.join(HERE, "image_w15.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; found =", found, "z =", z)


if __name__ == "__main__":
    main()

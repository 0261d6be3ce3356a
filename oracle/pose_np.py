"""
ORACLE (test infrastructure, not product code): scalar restatement of cv2.Rodrigues, the one
OpenCV routine the reference's pose recovery depends on
(/root/reference/keras_retinanet_3D/bin/run_network.py:178,188,...,327 `cv2.Rodrigues`).

OpenCV is a third-party dependency that is absent from /root/reference and from this image
(unpinned; README.md lists no version).  Restated from the OpenCV documentation of
cv::Rodrigues: matrix -> vector first replaces R by the nearest rotation (U V^T of its SVD), then
r = axis * angle with angle = acos((trace - 1) / 2); vector -> matrix is the Rodrigues formula
R = cos(t) I + (1 - cos(t)) k k^T + sin(t) [k]x.  "Parity unpinned" against OpenCV itself.

Used (a) as the `cv2.Rodrigues` stand-in when oracle/gen_harness_goldens.py executes the
reference's run_network.py, (b) by tests as the checker of utils.gpp_utils.
"""
import math

import numpy as np


def rodrigues(src):
    src = np.asarray(src, dtype=np.float64)
    if src.shape == (3, 3):
        U, _, Vt = np.linalg.svd(src)
        R = U.dot(Vt)
        rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
        s = math.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
        c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5, -1.0), 1.0)
        theta = math.acos(c)
        if s < 1e-5:
            if c > 0:
                r = np.zeros(3)
            else:
                rx = math.sqrt(max((R[0, 0] + 1) * 0.5, 0.0))
                ry = math.sqrt(max((R[1, 1] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
                rz = math.sqrt(max((R[2, 2] + 1) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
                if abs(rx) < abs(ry) and abs(rx) < abs(rz) and (R[1, 2] > 0) != (ry * rz > 0):
                    rz = -rz
                r = np.array([rx, ry, rz])
                r *= theta / np.linalg.norm(r)
        else:
            r = np.array([rx, ry, rz]) * (theta / (2.0 * s))
        return r.reshape(3, 1), None
    r = src.reshape(3)
    theta = float(np.linalg.norm(r))
    if theta < 1e-300:
        return np.eye(3), None
    k = r / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = math.cos(theta) * np.eye(3) + (1 - math.cos(theta)) * np.outer(k, k) + math.sin(theta) * K
    return R, None

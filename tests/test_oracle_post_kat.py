"""Known-answer tests of the CPU restatement of Frame::UndistortKeyPoints (cv::undistortPoints,
reference src/Frame.cc:421-451) and Frame::AssignFeaturesToGrid (src/Frame.cc:232-247, 398-408)."""
import numpy as np

K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)            # TUM1.yaml
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


def distort(xy, K, d):
    """Forward model of OpenCV (projectPoints), float64."""
    fx, fy, cx, cy = (float(v) for v in K)
    k1, k2, p1, p2, k3 = (float(v) for v in d)
    x, y = (xy[:, 0] - cx) / fx, (xy[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], 1)


def test_undistort_zero_k1_is_a_copy(oracle):
    xy = np.random.default_rng(0).uniform(0, 600, (50, 2)).astype(np.float32)
    out = oracle.undistort_points(xy, K, np.array([0, 0.3, 0.01, 0.02, 0.1], np.float32))
    assert (out.view(np.uint32) == xy.view(np.uint32)).all()


def test_undistort_principal_point_and_inverse(oracle):
    # the principal point is a fixed point of the model
    out = oracle.undistort_points(np.array([[K[2], K[3]]], np.float32), K, DIST)
    assert np.allclose(out, [[K[2], K[3]]], atol=1e-4)
    # undistort(distort(p)) ~ p: five iterations converge to well below a pixel for mild distortion
    mild = np.array([-0.12, 0.05, 0.001, -0.0007, 0.0], np.float32)
    rng = np.random.default_rng(1)
    ideal = np.stack([rng.uniform(20, 620, 500), rng.uniform(20, 460, 500)], 1)
    seen = distort(ideal, K, mild).astype(np.float32)
    back = oracle.undistort_points(seen, K, mild)
    assert np.abs(back - ideal).max() < 0.05


def test_undistort_matches_float64_numpy_restatement(oracle):
    rng = np.random.default_rng(2)
    xy = np.stack([rng.uniform(0, 640, 300), rng.uniform(0, 480, 300)], 1).astype(np.float32)
    fx, fy, cx, cy = (np.float64(v) for v in K)
    k1, k2, p1, p2, k3 = (np.float64(v) for v in DIST)
    x = (xy[:, 0].astype(np.float64) - cx) * (1.0 / fx)
    y = (xy[:, 1].astype(np.float64) - cy) * (1.0 / fy)
    x0, y0 = x.copy(), y.copy()
    for _ in range(5):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    exp = np.stack([fx * x + cx, fy * y + cy], 1).astype(np.float32)
    out = oracle.undistort_points(xy, K, DIST)
    assert (out.view(np.uint32) == exp.view(np.uint32)).all()


def test_grid_cells_and_order(oracle):
    # 640 x 480 image without distortion: inv_w = 64/640, inv_h = 48/480 (src/Frame.cc:101-102)
    inv_w, inv_h = np.float32(64) / np.float32(640), np.float32(48) / np.float32(480)
    xy = np.array([[0, 0], [4.9, 4.9], [5.1, 5.1], [634.9, 474.9], [635.1, 100], [100, 475.1], [14.9, 25.1],
                   [5.1, 5.2], [-6, 10]], np.float32)
    start, idx = oracle.assign_grid(xy, 0.0, 0.0, inv_w, inv_h)
    cells = {}
    for c in range(64 * 48):
        if start[c + 1] > start[c]:
            cells[(c // 48, c % 48)] = idx[start[c]:start[c + 1]].tolist()
    # round((x - 0) * 0.1): 0 -> 0, 4.9 -> 0, 5.1 -> 1, 634.9 -> 63, 635.1 -> 64 (outside), 475.1 -> 48 (outside),
    # 14.9 -> 1 / 25.1 -> 3, -6 -> -1 (outside; -0.6 rounds away from zero)
    assert cells == {(0, 0): [0, 1], (1, 1): [2, 7], (63, 47): [3], (1, 3): [6]}
    assert start[-1] == 6

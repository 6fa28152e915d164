"""Known-answer tests of the oracle's restatement of Frame::ComputeStereoFromRGBD (reference src/Frame.cc:662-682)
with the depth conversion of Tracking::GrabImageRGBD (src/Tracking.cc:230-231)."""
import numpy as np


def test_sampling_truncates_and_missing_depth_gives_minus_one(oracle):
    depth = np.zeros((4, 5), np.uint16)
    depth[3, 2] = 10000                      # 2 m at DepthMapFactor 5000
    depth[1, 4] = 2500
    xy = np.array([[2.9, 3.99], [4.0, 1.5], [0.2, 0.7], [2.0, 2.0]], np.float32)   # (x, y): (2,3), (4,1), (0,0), (2,2)
    un_x = np.array([2.5, 4.25, 0.0, 2.0], np.float32)
    ur, z = oracle.stereo_from_rgbd(depth, np.float32(1.0) / np.float32(5000.0), xy, un_x, 40.0)
    f = np.float32(1.0) / np.float32(5000.0)
    z0, z1 = np.float32(10000) * f, np.float32(2500) * f
    assert z[0] == z0 and z[1] == z1 and z[2] == -1 and z[3] == -1
    assert ur[0] == np.float32(2.5) - np.float32(40.0) / z0 and ur[1] == np.float32(4.25) - np.float32(40.0) / z1
    assert ur[2] == -1 and ur[3] == -1


def test_conversion_is_a_multiplication_by_the_float_factor(oracle):
    """convertTo(CV_32F, alpha) multiplies by (float)alpha; for some samples that differs from dividing by 5000."""
    f = np.float32(1.0) / np.float32(5000.0)
    raw = np.arange(1, 65536, dtype=np.uint16)
    mul = raw.astype(np.float32) * f
    div = raw.astype(np.float32) / np.float32(5000.0)
    differ = np.nonzero(mul != div)[0]
    assert len(differ) > 1000                 # the two forms are not the same function
    pick = raw[differ[:256]].reshape(16, 16)
    xy = np.stack(np.meshgrid(np.arange(16), np.arange(16)), -1).reshape(-1, 2).astype(np.float32) + np.float32(0.5)
    ur, z = oracle.stereo_from_rgbd(pick, f, xy, np.zeros(256, np.float32), 40.0)
    assert (z == pick.reshape(-1).astype(np.float32) * f).all()


def test_float_maps_and_non_positive_depths(oracle):
    depth = np.array([[0.0, -1.0, 0.5], [2.0, np.float32(1e-30), 3.0]], np.float32)
    xy = np.array([[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [2, 1]], np.float32)
    un_x = np.arange(6, dtype=np.float32)
    ur, z = oracle.stereo_from_rgbd(depth, 1.0, xy, un_x, 10.0)
    assert list(z[:2]) == [-1, -1] and z[2] == np.float32(0.5) and z[3] == 2 and z[5] == 3
    assert z[4] == np.float32(1e-30) and ur[4] == np.float32(4.0) - np.float32(10.0) / np.float32(1e-30)   # d > 0 is all that is asked
    assert ur[2] == np.float32(2.0) - np.float32(20.0) and ur[0] == -1
    # a factor on a float map (the reference converts when the factor is not 1, whatever the type)
    ur2, z2 = oracle.stereo_from_rgbd(depth, 0.5, xy, un_x, 10.0)
    assert z2[3] == 1.0 and ur2[3] == np.float32(3.0) - np.float32(10.0)

"""Known-answer tests of the oracle's windowed best-match search: Frame::GetFeaturesInArea
(reference src/Frame.cc:338-396) + the candidate loop of ORBmatcher::SearchByProjection
(reference src/ORBmatcher.cc:1427-1455)."""
import numpy as np


GRID = dict(min_x=0.0, min_y=0.0, inv_w=64 / 640.0, inv_h=48 / 480.0)   # 10 x 10 px cells


def setup(oracle, pts, octaves):
    xy = np.array(pts, np.float32)
    start, idx = oracle.assign_grid(xy, GRID["min_x"], GRID["min_y"], GRID["inv_w"], GRID["inv_h"])
    return xy, np.array(octaves, np.int32), start, idx


def search(oracle, xy, octv, desc, start, idx, u, v, r, lo=-1, hi=-1, ur=0.0, q=None, u_right=None, taken=None):
    q = np.zeros(32, np.uint8) if q is None else q
    return oracle.search_window(xy, octv, desc, u_right, taken, start, idx, GRID["min_x"], GRID["min_y"], GRID["inv_w"],
                                GRID["inv_h"], u, v, r, lo, hi, ur, q)


def test_window_is_open_and_levels_filter(oracle):
    xy, octv, start, idx = setup(oracle, [(100, 100), (110, 100), (100, 115), (300, 300)], [0, 1, 2, 0])
    desc = np.zeros((4, 32), np.uint8)
    desc[0, 0] = 0x07   # distance 3 to the zero query
    desc[1, 0] = 0x01   # distance 1
    desc[2, 0] = 0x03   # distance 2
    assert search(oracle, xy, octv, desc, start, idx, 100, 100, 10.5) == (1, 1)
    assert search(oracle, xy, octv, desc, start, idx, 100, 100, 10.0) == (0, 3)       # |dx| < r is strict: 110 is out
    assert search(oracle, xy, octv, desc, start, idx, 100, 100, 15.5) == (1, 1)
    assert search(oracle, xy, octv, desc, start, idx, 100, 100, 15.5, lo=2, hi=2) == (2, 2)
    assert search(oracle, xy, octv, desc, start, idx, 100, 100, 15.5, lo=0, hi=0) == (0, 3)
    assert search(oracle, xy, octv, desc, start, idx, 100, 100, 15.5, lo=1, hi=-1) == (1, 1)   # no upper limit
    assert search(oracle, xy, octv, desc, start, idx, 500, 100, 20.0) == (-1, 256)             # empty window
    assert search(oracle, xy, octv, desc, start, idx, -500, 100, 20.0) == (-1, 256)            # window left of the grid
    assert search(oracle, xy, octv, desc, start, idx, 5000, 100, 20.0) == (-1, 256)            # right of it


def test_ties_follow_grid_order(oracle):
    # equal distances: the first candidate in GetFeaturesInArea's order wins -- cells column by column
    # (ix outer, iy inner), features of a cell in index order
    pts = [(125, 105), (105, 125), (105, 105), (106, 106)]
    xy, octv, start, idx = setup(oracle, pts, [0, 0, 0, 0])
    desc = np.zeros((4, 32), np.uint8)
    assert search(oracle, xy, octv, desc, start, idx, 115, 115, 30.0) == (2, 0)   # cell (10,10) before (10,12), (12,10)
    desc[2, 0] = 1
    assert search(oracle, xy, octv, desc, start, idx, 115, 115, 30.0) == (3, 0)   # then index 3 of the same cell
    desc[3, 0] = 1
    assert search(oracle, xy, octv, desc, start, idx, 115, 115, 30.0) == (1, 0)   # column 10 row 12 before column 12
    full = np.full(32, 0xFF, np.uint8)
    assert search(oracle, xy, octv, np.zeros((4, 32), np.uint8), start, idx, 115, 115, 30.0, q=full) == (-1, 256)  # 256 is not < 256


def test_taken_and_right_coordinate(oracle):
    xy, octv, start, idx = setup(oracle, [(200, 200), (203, 200), (206, 200)], [0, 0, 0])
    desc = np.zeros((3, 32), np.uint8)
    desc[1, 0] = 1
    desc[2, 0] = 3
    taken = np.array([1, 0, 0], np.uint8)
    assert search(oracle, xy, octv, desc, start, idx, 202, 200, 8.0, taken=taken) == (1, 1)
    u_right = np.array([-1.0, 150.0, 190.0], np.float32)      # feature 0 has none: no check
    assert search(oracle, xy, octv, desc, start, idx, 202, 200, 8.0, ur=192.0, u_right=u_right) == (0, 0)
    assert search(oracle, xy, octv, desc, start, idx, 202, 200, 8.0, ur=192.0, u_right=u_right, taken=taken) == (2, 2)
    assert search(oracle, xy, octv, desc, start, idx, 202, 200, 8.0, ur=158.0, u_right=u_right, taken=taken) == (1, 1)  # er == r passes

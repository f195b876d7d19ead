"""KATs of the frame-ingest restatement (oracle/ingest_oracle.py; SURVEY §8f rank 3, Frame.cpp:45-75)."""
import numpy as np
from oracle import ingest_oracle as I

REF_K = np.array([1642.405612, 1636.148027, 960.0, 540.0], np.float32)                  # ExternVariable.h:53-59 x INTRINSIC_FACTOR
REF_DIST = np.array([-0.288283, 0.146546, 0.003800, -0.001690, -0.132134], np.float32)    # ExternVariable.h:62


def test_gray_is_the_fixed_point_luma():
    assert I.bgr2gray(np.array([[[255, 255, 255]]], np.uint8))[0, 0] == 255     # coefficients sum to 2^14
    assert I.bgr2gray(np.array([[[0, 0, 0]]], np.uint8))[0, 0] == 0
    assert I.bgr2gray(np.array([[[255, 0, 0]]], np.uint8))[0, 0] == (1868 * 255 + 8192) >> 14    # blue is channel 0
    assert I.bgr2gray(np.array([[[0, 0, 255]]], np.uint8))[0, 0] == (4899 * 255 + 8192) >> 14
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    ref = np.floor((0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]) + 0.5)
    assert np.abs(I.bgr2gray(img).astype(int) - ref).max() <= 1


def test_inverse_of_a_camera_matrix():
    ir = I.inv3x3([800.0, 0, 320.5, 0, 790.0, 240.25, 0, 0, 1]).reshape(3, 3)
    assert np.allclose(ir @ np.array([[800.0, 0, 320.5], [0, 790.0, 240.25], [0, 0, 1]]), np.eye(3), atol=1e-12)
    assert np.all(I.inv3x3(np.zeros(9)) == 0)


def test_zero_distortion_gives_the_identity_pipeline():
    rng = np.random.default_rng(2)
    w, h = 64, 48
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    K = np.array([50.0, 52.0, 31.0, 23.0], np.float32)
    Kn = I.optimal_new_camera(K, np.zeros(5), w, h)
    # without distortion the grid maps linearly: the inscribed rectangle is the image, new K reproduces x -> x*(w-1)/w
    assert abs(Kn[0] - K[0] * (w - 1) / w) < 1e-3 and abs(Kn[1] - K[1] * (h - 1) / h) < 1e-3
    ix, iy, fx5, fy5 = I.undistort_maps(K, np.zeros(5), K, w, h)        # same camera in and out: exact identity map
    assert np.array_equal(ix, np.tile(np.arange(w), (h, 1))) and np.array_equal(iy, np.tile(np.arange(h)[:, None], (1, w)))
    assert not fx5.any() and not fy5.any()
    gray = I.bgr2gray(bgr)
    assert np.array_equal(I.remap_bilinear(gray, ix, iy, fx5, fy5), gray)
    img, g2, und, _ = I.ingest(bgr, K, np.zeros(5), do_undistort=False)
    assert np.array_equal(g2, gray) and img.shape == (h // 4, w // 4)
    a = gray.astype(int)
    assert img[3, 5] == (a[13, 21] + a[13, 22] + a[14, 21] + a[14, 22] + 2) >> 2


def test_remap_weights_and_border():
    gray = np.array([[0, 64], [128, 255]], np.uint8)
    one = lambda v: np.array([[v]], np.int32)
    # centre of the four pixels: fractions 16/32 each => plain average, rounded half up by the +2^14
    assert I.remap_bilinear(gray, one(0), one(0), one(16), one(16))[0, 0] == (0 + 64 + 128 + 255 + 2) // 4
    # a tap row outside the image contributes the border constant 0
    assert I.remap_bilinear(gray, one(0), one(1), one(0), one(16))[0, 0] == (128 * 16 * 32 * 32 + (1 << 14)) >> 15
    assert I.remap_bilinear(gray, one(5), one(5), one(3), one(7))[0, 0] == 0


def test_reference_camera_new_matrix_and_map_sanity():
    w, h = 1920, 1080
    Kn = I.optimal_new_camera(REF_K, REF_DIST, w, h)
    # barrel distortion (k1 < 0): alpha = 0 zooms in, the principal point stays near the centre
    assert Kn[0] > 0.5 * REF_K[0] and Kn[1] > 0.5 * REF_K[1] and abs(Kn[2] - 960) < 40 and abs(Kn[3] - 540) < 40
    ix, iy, fx5, fy5 = I.undistort_maps(REF_K, REF_DIST, Kn, w, h)
    # alpha = 0: every output pixel samples inside the source image
    assert ix.min() >= -1 and ix.max() <= w and iy.min() >= -1 and iy.max() <= h
    u = ix + fx5 / 32.0; v = iy + fy5 / 32.0
    # the map is smooth and orientation preserving
    assert (np.diff(u, axis=1) > 0).all() and (np.diff(v, axis=0) > 0).all()
    # the principal point of the new camera maps to the principal point of the old one (no distortion at r = 0)
    cxn, cyn = int(round(float(Kn[2]))), int(round(float(Kn[3])))
    assert abs(u[cyn, cxn] - 960) < 1.5 and abs(v[cyn, cxn] - 540) < 1.5


def test_golden_ingest_small_regression():
    """tests/golden/ingest_small.npz: pins the ingest restatement (generator: tests/golden/make_golden.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as G
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ingest_small.npz"))
    img, gray, und, kn = I.ingest(G.ingest_small_frame(), G.INGEST_K, G.INGEST_DIST, True)
    assert np.array_equal(img, g["image"]) and np.array_equal(gray, g["gray"]) and np.array_equal(und, g["undistorted"])
    assert np.array_equal(kn.view(np.uint32), g["new_camera"].view(np.uint32))
    assert und.min() < und.max() and (und == 0).mean() < 0.01     # alpha = 0: no black border

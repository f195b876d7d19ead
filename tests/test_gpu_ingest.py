"""Frame ingest on the device (ellc_ingest_configure / ellc_frame_ingest_bgr) against the numpy restatement:
integer pipeline => every byte equal. Reference: frame::frame(VideoCapture), Frame.cpp:45-75."""
import numpy as np
import pytest
from oracle import ingest_oracle as I

pytestmark = pytest.mark.gpu

REF_K = np.array([1642.405612, 1636.148027, 960.0, 540.0], np.float32)                  # ExternVariable.h:53-59 x INTRINSIC_FACTOR
REF_DIST = np.array([-0.288283, 0.146546, 0.003800, -0.001690, -0.132134], np.float32)    # ExternVariable.h:62


def frames(w, h, seed):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) % 256)], -1).astype(np.uint8)
    return [noise, smooth]


@pytest.mark.parametrize("w,h,K,dist", [(1920, 1080, REF_K, REF_DIST),
                                         (640, 480, np.array([420.0, 415.0, 322.0, 236.0], np.float32),
                                          np.array([0.11, -0.05, -0.002, 0.001, 0.01], np.float32))])
def test_ingest_matches_the_restatement_byte_for_byte(ellc, w, h, K, dist):
    ctx = ellc.Context(ellc.default_config(w // 4, h // 4, 4, max_frames=2))
    kn = ctx.ingest_configure(w, h, K[0], K[1], K[2], K[3], dist, True)
    assert np.array_equal(kn.view(np.uint32), I.optimal_new_camera(K, dist, w, h).view(np.uint32))
    for bgr in frames(w, h, 5):
        gray_p, und_p = ctx.frame_ingest_bgr(1, bgr, probes=True)
        img, gray, und, _ = I.ingest(bgr, K, dist, True)
        assert np.array_equal(gray_p, gray[1::4, 1::4])
        ref4 = np.stack([und[1::4, 1::4], und[1::4, 2::4], und[2::4, 1::4], und[2::4, 2::4]], -1)
        assert np.array_equal(und_p, ref4)
        lvl0, (rows, cols) = ctx.image_level(False, 1, 0)
        assert (cols, rows) == (w // 4, h // 4) and np.array_equal(lvl0[:rows, :cols], img)
        # the pyramid above it is the ordinary one: same as uploading the ingested image
        ctx.frame_upload(0, img)
        for l in range(1, 4):
            a = ctx.image_level(False, 1, l)[0]; b = ctx.image_level(False, 0, l)[0]
            assert np.array_equal(a, b)
    ctx.close()


def test_ingest_without_undistortion_and_errors(ellc):
    w, h = 256, 192
    ctx = ellc.Context(ellc.default_config(w // 4, h // 4, 3, max_frames=1))
    with pytest.raises(ellc.EllcError):
        ctx.frame_ingest_bgr(0, np.zeros((h, w, 3), np.uint8))             # not configured
    with pytest.raises(ellc.EllcError):
        ctx.ingest_configure(w + 4, h, 100.0, 100.0, 64.0, 48.0, None, False)   # context is not input / 4
    ctx.ingest_configure(w, h, 200.0, 200.0, 128.0, 96.0, None, False)
    bgr = frames(w, h, 9)[0]
    ctx.frame_ingest_bgr(0, bgr)
    img = I.ingest(bgr, np.array([200, 200, 128, 96], np.float32), np.zeros(5), False)[0]
    got = ctx.image_level(False, 0, 0)[0]
    assert np.array_equal(got[: h // 4, : w // 4], img)
    ctx.close()

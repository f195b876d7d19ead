"""CPU restatement of the reference's frame ingest after the decode (frame::frame(VideoCapture), Frame.cpp:45-75):
cvtColor BGR2GRAY -> getOptimalNewCameraMatrix(alpha=0) -> undistort -> resize(1/4, INTER_LINEAR).

TEST INFRASTRUCTURE ONLY (see oracle/README in ellc_oracle.hpp): imported by tests/, never by the product.
PARITY UNPINNED: OpenCV 3.0.0 is neither in /root/reference nor installed; the functions below restate its published
algorithms (modules/imgproc/src/{color,undistort,imgwarp}.cpp, modules/calib3d/src/calibration.cpp) in numpy:
  gray      (1868 B + 9617 G + 4899 R + 8192) >> 14                                     [RGB2Gray<uchar>, yuv_shift 14]
  new K     9x9 grid -> cvUndistortPoints (5 iterations, double, stored f32) -> inscribed rectangle -> fx0 = (w-1)/inner.w
            (f32 division), cx0 = -fx0*inner.x; result stored in the camera matrix' type (f32)   [cvGetOptimalNewCameraMatrix]
  undistort stripes of max(1, 4096/w) rows, per stripe inv(3x3, closed form) of the new K with cy-y0, per row running
            sums _x += ir[0] (np.add.accumulate = the same left fold), model in double, coordinates rounded half-to-even
            to 1/32 px                                                                        [undistort, initUndistortRectifyMap]
  remap     bilinear 8u: weights (32-fy)(32-fx)*32 ..., (sum + 2^14) >> 15, BORDER_CONSTANT 0      [remapBilinear, BilinearTab_i]
  resize    scale 4, INTER_LINEAR 8u: columns 4dx+1, 4dx+2 with weights 1024/1024 (rows likewise);
            ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2 = (p00+p01+p10+p11+2)>>2             [resizeGeneric_, VResizeLinear]
"""
import numpy as np

F32_MAX = np.float32(3.402823466e+38)


def bgr2gray(bgr):
    b = bgr[..., 0].astype(np.int64); g = bgr[..., 1].astype(np.int64); r = bgr[..., 2].astype(np.int64)
    return ((1868 * b + 9617 * g + 4899 * r + 8192) >> 14).astype(np.uint8)


def inv3x3(S):
    S = np.asarray(S, np.float64).reshape(9)
    d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6])
    if d0 == 0.0:
        return np.zeros(9)
    d = 1.0 / d0
    return np.array([(S[4] * S[8] - S[5] * S[7]) * d, (S[2] * S[7] - S[1] * S[8]) * d, (S[1] * S[5] - S[2] * S[4]) * d,
                     (S[5] * S[6] - S[3] * S[8]) * d, (S[0] * S[8] - S[2] * S[6]) * d, (S[2] * S[3] - S[0] * S[5]) * d,
                     (S[3] * S[7] - S[4] * S[6]) * d, (S[1] * S[6] - S[0] * S[7]) * d, (S[0] * S[4] - S[1] * S[3]) * d])


def optimal_new_camera(K4, dist5, w, h):
    """K4 = f32 (fx, fy, cx, cy); returns the f32 (fx, fy, cx, cy) of getOptimalNewCameraMatrix(alpha=0)."""
    K4 = np.asarray(K4, np.float32); k = np.asarray(dist5, np.float32).astype(np.float64)
    fx, fy, cx, cy = [float(v) for v in K4]
    ifx, ify = 1.0 / fx, 1.0 / fy
    N = 9
    iX0, iX1, iY0, iY1 = -F32_MAX, F32_MAX, -F32_MAX, F32_MAX
    for yy in range(N):
        for xx in range(N):
            px = np.float32(np.float32(np.float32(xx) * np.float32(w)) / np.float32(N - 1))
            py = np.float32(np.float32(np.float32(yy) * np.float32(h)) / np.float32(N - 1))
            x = (float(px) - cx) * ifx; y = (float(py) - cy) * ify
            x0, y0 = x, y
            for _ in range(5):
                r2 = x * x + y * y
                icdist = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
                dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
                dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
                x = (x0 - dX) * icdist; y = (y0 - dY) * icdist
            ux, uy = np.float32(x), np.float32(y)
            if xx == 0: iX0 = max(iX0, ux)
            if xx == N - 1: iX1 = min(iX1, ux)
            if yy == 0: iY0 = max(iY0, uy)
            if yy == N - 1: iY1 = min(iY1, uy)
    iw = np.float32(iX1 - iX0); ih = np.float32(iY1 - iY0)
    fx0 = float(np.float32(w - 1) / iw); fy0 = float(np.float32(h - 1) / ih)
    cx0 = -fx0 * float(iX0); cy0 = -fy0 * float(iY0)
    return np.array([fx0, fy0, cx0, cy0], np.float32)


def undistort_maps(K4, dist5, Knew4, w, h):
    """Fixed-point maps of cv::undistort: ix, iy (int16 semantics) and the 5+5-bit fraction (fy5, fx5) per pixel."""
    K4 = np.asarray(K4, np.float32).astype(np.float64); Kn = np.asarray(Knew4, np.float32).astype(np.float64)
    k1, k2, p1, p2, k3 = np.asarray(dist5, np.float32).astype(np.float64)
    fx, fy, u0, v0 = K4
    ix = np.zeros((h, w), np.int32); iy = np.zeros((h, w), np.int32); fx5 = np.zeros((h, w), np.int32); fy5 = np.zeros((h, w), np.int32)
    stripe0 = min(max(1, (1 << 12) // max(w, 1)), h)
    for y0 in range(0, h, stripe0):
        ir = inv3x3([Kn[0], 0, Kn[2], 0, Kn[1], Kn[3] - y0, 0, 0, 1])
        for i in range(min(stripe0, h - y0)):
            def run(start, step):   # _x at column j = ((start + step) + step) ... : left fold, like the reference's +=
                a = np.full(w, step, np.float64); a[0] = start
                return np.add.accumulate(a)
            _x = run(i * ir[1] + ir[2], ir[0]); _y = run(i * ir[4] + ir[5], ir[3]); _w = run(i * ir[7] + ir[8], ir[6])
            ww = 1.0 / _w; x = _x * ww; y = _y * ww
            x2 = x * x; y2 = y * y; r2 = x2 + y2; _2xy = 2 * x * y
            kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / 1.0
            u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0
            v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0
            iu = np.rint(u * 32).astype(np.int64); iv = np.rint(v * 32).astype(np.int64)
            ix[y0 + i] = iu >> 5; iy[y0 + i] = iv >> 5; fx5[y0 + i] = iu & 31; fy5[y0 + i] = iv & 31
    return ix, iy, fx5, fy5


def remap_bilinear(gray, ix, iy, fx5, fy5):
    h, w = gray.shape
    g = np.zeros((h + 2, w + 2), np.int64)          # one-pixel zero frame = BORDER_CONSTANT(0) for partially outside taps
    g[1:-1, 1:-1] = gray

    def tap(xx, yy):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, g[np.clip(yy, -1, h) + 1, np.clip(xx, -1, w) + 1], 0)
    w00 = (32 - fy5) * (32 - fx5) * 32; w01 = (32 - fy5) * fx5 * 32; w10 = fy5 * (32 - fx5) * 32; w11 = fy5 * fx5 * 32
    s = tap(ix, iy) * w00 + tap(ix + 1, iy) * w01 + tap(ix, iy + 1) * w10 + tap(ix + 1, iy + 1) * w11
    return ((s + (1 << 14)) >> 15).astype(np.uint8)


def resize_quarter(img):
    a = img.astype(np.int64)
    return ((a[1::4, 1::4] + a[1::4, 2::4] + a[2::4, 1::4] + a[2::4, 2::4] + 2) >> 2).astype(np.uint8)


def ingest(bgr, K4, dist5, do_undistort=True):
    """Returns (image W/4 x H/4, gray full size, undistorted full size, new camera f32[4])."""
    h, w = bgr.shape[:2]
    gray = bgr2gray(bgr)
    if not do_undistort:
        return resize_quarter(gray), gray, gray, np.asarray(K4, np.float32)
    Kn = optimal_new_camera(K4, dist5, w, h)
    ix, iy, fx5, fy5 = undistort_maps(K4, dist5, Kn, w, h)
    und = remap_bilinear(gray, ix, iy, fx5, fy5)
    return resize_quarter(und), gray, und, Kn

"""SECOND SOURCE of the depth side of the oracle (test infrastructure; r05 verdict, "second-source the depth oracle").

A scalar f32 restatement, in numpy scalars, of

    depthMap::observeDepthRow / observeDepthCreate        DepthPropagation.cpp:191-308
    depthMap::makeAndCheckEPL                             DepthPropagation.cpp:311-384
    depthMap::doLineStereo                                DepthPropagation.cpp:397-885
    depthMap::observeDepthUpdate                          DepthPropagation.cpp:888-999
    depthMap::propagateDepth                              DepthPropagation.cpp:1003-1157
    frame::getInterpolatedElement (u8 and gradient)       Frame.h:181-394
    frame::calculateGradient                              Frame.cpp:185-285

written FROM THE REFERENCE'S TEXT, statement by statement, WITHOUT looking at oracle/ellc_oracle_depth.cpp or at the HIP kernels: the
C++ oracle and the kernels are one author's reading of those lines, and a shared misreading would be green in every GPU-vs-oracle
test. tests/test_second_source_depth.py runs both on the same scenes and compares every field of every pixel.

Taken as given (third-party arithmetic, pinned elsewhere or unpinnable here): the frame-to-frame matrices of Frame.cpp:376-413 (Eigen
exp / inverse), ORIG_*_INV (cv::Mat::inv), maxAbsGradient (Frame.cpp:618-674, its own numpy twin is synth.max_abs_gradient).

Every operation is on numpy float32 scalars in the reference's expression order (x86-64, no contraction); where the reference's types
promote (int * float, double literals in UNZERO, int counters updated with float constants) the promotion is written out. Eigen's
fixed-size products are coefficient-wise left to right; Eigen's `a.dot(b)` on fixed 3-vectors is a REDUX, which Eigen 3.2 unrolls as a
binary tree: a0 b0 + (a1 b1 + a2 b2) (Redux.h, redux_novec_unroller) — see `EIGEN_DOT_TREE` below."""
import math
import numpy as np

F = np.float32
I32 = np.int32

# ExternVariable.h:80-140
MIN_ABS_GRAD_CREATE = F(1.0)
MIN_ABS_GRAD_DECREASE = F(5.0)
MIN_BLACKLIST = -1
MAX_DIFF_CONSTANT = F(40.0) * F(40.0)
MAX_DIFF_GRAD_MULT = F(0.5) * F(0.5)
MIN_EPL_GRAD_SQUARED = F(2.0) * F(2.0)
MIN_EPL_LENGTH_SQUARED = F(1.0) * F(1.0)
MIN_EPL_ANGLE_SQUARED = F(0.3) * F(0.3)
MIN_DEPTH = F(0.05)
MAX_EPL_LENGTH_CROP = F(30.0)
MIN_EPL_LENGTH_CROP = F(3.0)
GRADIENT_SAMPLE_DIST = F(1.0)
SAMPLE_POINT_TO_BORDER = F(7.0)
MAX_ERROR_STEREO = F(1300.0)
MIN_DISTANCE_ERROR_STEREO = F(1.5)
STEREO_EPL_VAR_FAC = F(2.0)
DIVISION_EPS = F(1e-10)
CAMERA_PIXEL_NOISE = 4 * 4                 # static const int
VALIDITY_COUNTER_INITIAL_OBSERVE = 5       # int
SUCC_VAR_INC_FAC = F(1.01)
FAIL_VAR_INC_FAC = F(1.1)
MAX_VAR = F(0.5) * F(0.5)
DIFF_FAC_OBSERVE = F(1.0) * F(1.0)
DIFF_FAC_PROP_MERGE = F(1.0) * F(1.0)
VALIDITY_COUNTER_MAX = F(5.0)
VALIDITY_COUNTER_MAX_VARIABLE = F(250.0)
VALIDITY_COUNTER_DEC = F(5.0)
VALIDITY_COUNTER_INC = F(5.0)

# Eigen 3.2 unrolls the sum of a fixed-size redux as a binary tree: for three terms e0 + (e1 + e2). False: (e0 + e1) + e2.
EIGEN_DOT_TREE = True


def unzero(val):
    """#define UNZERO(val) (val < 0 ? (val > -1e-10 ? -1e-10 : val) : (val < 1e-10 ? 1e-10 : val)) — double literals: the comparisons
    and the clamped value are double, the result is assigned to a float."""
    v = np.float64(val)
    if v < 0:
        r = np.float64(-1e-10) if v > -1e-10 else v
    else:
        r = np.float64(1e-10) if v < 1e-10 else v
    return F(r)


def c_int(x):
    """C conversion float -> int (truncation toward zero)"""
    return int(math.trunc(float(x)))


def calculate_gradient(img):
    """frame::calculateGradient (Frame.cpp:185-285) at level 0: 0.5 * central difference inside, one-sided difference (no 0.5) on the
    border row / column of the respective direction."""
    f = img.astype(np.float32)
    rows, cols = f.shape
    gx = np.zeros((rows, cols), np.float32)
    gy = np.zeros((rows, cols), np.float32)
    # CASE 1: interior
    gx[1:-1, 1:-1] = F(0.5) * (f[1:-1, 2:] - f[1:-1, :-2])
    gy[1:-1, 1:-1] = F(0.5) * (f[2:, 1:-1] - f[:-2, 1:-1])
    # CASE 2: top row
    y = 0
    gx[y, 0] = f[y, 1] - f[y, 0]
    gy[y, 0] = f[y + 1, 0] - f[y, 0]
    gx[y, 1:-1] = F(0.5) * (f[y, 2:] - f[y, :-2])
    gy[y, 1:-1] = f[y + 1, 1:-1] - f[y, 1:-1]
    gx[y, cols - 1] = f[y, cols - 1] - f[y, cols - 2]
    gy[y, cols - 1] = f[y + 1, cols - 1] - f[y, cols - 1]
    # CASE 3: bottom row
    y = rows - 1
    gx[y, 0] = f[y, 1] - f[y, 0]
    gy[y, 0] = f[y, 0] - f[y - 1, 0]
    gx[y, 1:-1] = F(0.5) * (f[y, 2:] - f[y, :-2])
    gy[y, 1:-1] = f[y, 1:-1] - f[y - 1, 1:-1]
    gx[y, cols - 1] = f[y, cols - 1] - f[y, cols - 2]
    gy[y, cols - 1] = f[y, cols - 1] - f[y - 1, cols - 1]
    # CASE 4: left and right border
    gx[1:-1, 0] = f[1:-1, 1] - f[1:-1, 0]
    gy[1:-1, 0] = F(0.5) * (f[2:, 0] - f[:-2, 0])
    gx[1:-1, cols - 1] = f[1:-1, cols - 1] - f[1:-1, cols - 2]
    gy[1:-1, cols - 1] = F(0.5) * (f[2:, cols - 1] - f[:-2, cols - 1])
    return gx, gy


def interp(img, x1, y1):
    """frame::getInterpolatedElement (Frame.h:181-279 for the u8 image, :283-394 for a gradient plane; checkOutfBound = 0): the four
    taps are (floor x, floor y), (CEIL x, floor y), (floor x, CEIL y), (ceil x, ceil y) — ceil, not floor + 1 — each tested against the
    image with ITS OWN pair of coordinates: the `ceil` taps are tested with the unrounded x1 / y1."""
    rows, cols = img.shape
    nCols = cols - 1
    nRows = rows - 1
    x1 = F(x1); y1 = F(y1)
    wt0 = y1 - F(np.floor(y1))
    wt1 = x1 - F(np.floor(x1))
    # Case 1
    y = F(np.floor(y1)); x = F(np.floor(x1))
    if (x < 0) or (x > nCols) or (y < 0) or (y > nRows):
        pix1 = F(0)
    else:
        pix1 = F(img[int(y), int(x)])
    x = x1
    if (x < 0) or (x > nCols) or (y < 0) or (y > nRows):
        pix2 = F(0)
    else:
        pix2 = F(img[int(y), int(np.ceil(x))])
    interTop = ((F(1) - wt1) * pix1) + (wt1 * pix2)
    # Case 2
    y = y1
    x = F(np.floor(x1))
    if (x < 0) or (x > nCols) or (y < 0) or (y > nRows):
        pix1 = F(0)
    else:
        pix1 = F(img[int(np.ceil(y)), int(x)])
    x = x1
    if (x < 0) or (x > nCols) or (y < 0) or (y > nRows):
        pix2 = F(0)
    else:
        pix2 = F(img[int(np.ceil(y)), int(np.ceil(x))])
    interBtm = ((F(1) - wt1) * pix1) + (wt1 * pix2)
    return ((F(1) - wt0) * interTop) + (wt0 * interBtm)


def dot3_fixed(a, b):
    """Eigen: Vector3f.dot(row of a Matrix3f) — a fixed-size redux"""
    p0 = F(a[0]) * F(b[0]); p1 = F(a[1]) * F(b[1]); p2 = F(a[2]) * F(b[2])
    return p0 + (p1 + p2) if EIGEN_DOT_TREE else (p0 + p1) + p2


def matvec3(M, v):
    """Eigen: Matrix3f * Vector3f — coefficient-based product, each coefficient accumulated left to right"""
    return [(F(M[r][0]) * v[0] + F(M[r][1]) * v[1]) + F(M[r][2]) * v[2] for r in range(3)]


class DepthSecondSource:
    """state: dict of (H, W) arrays invDepth, invDepthSmoothed, variance, varianceSmoothed (f32), validity, blacklisted (i32), valid (u8)"""

    def __init__(self, W, H, fx, fy, cx, cy, kinv, kf_image, kf_maxgrad, state):
        self.W, self.H = W, H
        self.FX, self.FY, self.CX, self.CY = F(fx), F(fy), F(cx), F(cy)
        self.FX_INV, self.FY_INV, self.CX_INV, self.CY_INV = (F(v) for v in kinv)
        self.kf_image = kf_image
        self.kf_gradx, self.kf_grady = calculate_gradient(kf_image)
        self.kf_maxgrad = kf_maxgrad
        self.st = {k: np.array(v, copy=True) for k, v in state.items()}
        self.stereo_returns = {}     # histogram of doLineStereo's return classes (-1 .. -4, 0 = success)
        self.update_returns = {}     # histogram of observeDepthUpdate's return values

    # ---- the current frame and its matrices (Frame.cpp:376-413, taken as given)
    def set_current(self, cur_image, mats):
        self.cur_image = cur_image
        self.OtherWrtThis_t = [F(mats["OtherWrtThis"][r, 3]) for r in range(3)]
        self.ThisWrtOther_r = [[F(mats["ThisWrtOther"][r, c]) for c in range(3)] for r in range(3)]
        self.ThisWrtOther_t = [F(mats["ThisWrtOther"][r, 3]) for r in range(3)]
        self.K_r = [[F(mats["K_r"][r, c]) for c in range(3)] for r in range(3)]
        self.K_t = [F(mats["K_t"][r]) for r in range(3)]

    # ---- DepthPropagation.cpp:311-384
    def make_and_check_epl(self, x, y):
        t = self.OtherWrtThis_t
        epx = -self.FX * t[0] + t[2] * (F(x) - self.CX)
        epy = -self.FY * t[1] + t[2] * (F(y) - self.CY)
        if np.isnan(epx + epy):
            return None
        eplLengthSquared = epx * epx + epy * epy
        if eplLengthSquared < MIN_EPL_LENGTH_SQUARED:
            return None
        img = self.kf_image
        gx = F(int(img[y, x + 1]) - int(img[y, x - 1]))     # uchar - uchar: int arithmetic, then to float
        gy = F(int(img[y + 1, x]) - int(img[y - 1, x]))
        eplGradSquared = gx * epx + gy * epy
        eplGradSquared = eplGradSquared * eplGradSquared / eplLengthSquared
        if eplGradSquared < MIN_EPL_GRAD_SQUARED:
            return None
        if eplGradSquared / (gx * gx + gy * gy) < MIN_EPL_ANGLE_SQUARED:
            return None
        fac = GRADIENT_SAMPLE_DIST / F(np.sqrt(eplLengthSquared))
        return epx * fac, epy * fac

    # ---- DepthPropagation.cpp:397-885. Returns (error, result_idepth, result_var, result_eplLength)
    def do_line_stereo(self, u, v, epxn, epyn, min_idepth, prior_idepth, max_idepth):
        u = F(u); v = F(v); epxn = F(epxn); epyn = F(epyn)
        min_idepth = F(min_idepth); prior_idepth = F(prior_idepth); max_idepth = F(max_idepth)
        W, H = self.W, self.H
        kf, cur = self.kf_image, self.cur_image
        NAN = F(np.nan)
        KinvP = [self.FX_INV * u + self.CX_INV, self.FY_INV * v + self.CY_INV, F(1.0)]
        pInf = matvec3(self.K_r, KinvP)
        pReal = [pInf[i] / prior_idepth + self.K_t[i] for i in range(3)]
        rescaleFactor = pReal[2] * prior_idepth
        firstX = u - F(2) * epxn * rescaleFactor
        firstY = v - F(2) * epyn * rescaleFactor
        lastX = u + F(2) * epxn * rescaleFactor
        lastY = v + F(2) * epyn * rescaleFactor
        # (the comparisons are against ints converted to float)
        if (firstX <= 0 or firstX >= F(W - 2) or firstY <= 0 or firstY >= F(H - 2) or lastX <= 0 or lastX >= F(W - 2) or lastY <= 0 or lastY >= F(H - 2)):
            return F(-1), NAN, NAN, NAN
        if not (rescaleFactor > F(0.7) and rescaleFactor < F(1.4)):
            return F(-1), NAN, NAN, NAN
        realVal_p1 = interp(kf, u + epxn * rescaleFactor, v + epyn * rescaleFactor)
        realVal_m1 = interp(kf, u - epxn * rescaleFactor, v - epyn * rescaleFactor)
        realVal = interp(kf, u, v)
        realVal_m2 = interp(kf, u - F(2) * epxn * rescaleFactor, v - F(2) * epyn * rescaleFactor)
        realVal_p2 = interp(kf, u + F(2) * epxn * rescaleFactor, v + F(2) * epyn * rescaleFactor)
        pClose = [pInf[i] + self.K_t[i] * max_idepth for i in range(3)]
        if pClose[2] < F(0.001):
            max_idepth = (F(0.001) - pInf[2]) / self.K_t[2]
            pClose = [pInf[i] + self.K_t[i] * max_idepth for i in range(3)]
        pc2 = pClose[2]
        pClose = [pClose[i] / pc2 for i in range(3)]
        pFar = [pInf[i] + self.K_t[i] * min_idepth for i in range(3)]
        if pFar[2] < F(0.001) or max_idepth < min_idepth:
            return F(-1), NAN, NAN, NAN
        pf2 = pFar[2]
        pFar = [pFar[i] / pf2 for i in range(3)]
        if np.isnan(pFar[0] + pClose[0]):
            return F(-4), NAN, NAN, NAN
        incx = pClose[0] - pFar[0]
        incy = pClose[1] - pFar[1]
        eplLength = F(np.sqrt(incx * incx + incy * incy))
        # `if(!eplLength > 0 || std::isinf(eplLength))`: (!eplLength) > 0, i.e. eplLength == 0 — a NaN is NOT caught here
        if (1 if eplLength == 0 else 0) > 0 or np.isinf(eplLength):
            return F(-4), NAN, NAN, NAN
        if eplLength > MAX_EPL_LENGTH_CROP:
            pClose[0] = pFar[0] + incx * MAX_EPL_LENGTH_CROP / eplLength
            pClose[1] = pFar[1] + incy * MAX_EPL_LENGTH_CROP / eplLength
        incx = incx * (GRADIENT_SAMPLE_DIST / eplLength)
        incy = incy * (GRADIENT_SAMPLE_DIST / eplLength)
        pFar[0] = pFar[0] - incx
        pFar[1] = pFar[1] - incy
        pClose[0] = pClose[0] + incx
        pClose[1] = pClose[1] + incy
        if eplLength < MIN_EPL_LENGTH_CROP:
            pad = (MIN_EPL_LENGTH_CROP - eplLength) / F(2.0)
            pFar[0] = pFar[0] - incx * pad
            pFar[1] = pFar[1] - incy * pad
            pClose[0] = pClose[0] + incx * pad
            pClose[1] = pClose[1] + incy * pad
        B = SAMPLE_POINT_TO_BORDER
        colsB = F(W) - B      # util::ORIG_COLS - SAMPLE_POINT_TO_BORDER: int - float
        rowsB = F(H) - B
        if pFar[0] <= B or pFar[0] >= colsB or pFar[1] <= B or pFar[1] >= rowsB:
            return F(-1), NAN, NAN, NAN
        if pClose[0] <= B or pClose[0] >= colsB or pClose[1] <= B or pClose[1] >= rowsB:
            if pClose[0] <= B:
                toAdd = (B - pClose[0]) / incx
                pClose[0] = pClose[0] + toAdd * incx
                pClose[1] = pClose[1] + toAdd * incy
            elif pClose[0] >= colsB:
                toAdd = (colsB - pClose[0]) / incx
                pClose[0] = pClose[0] + toAdd * incx
                pClose[1] = pClose[1] + toAdd * incy
            if pClose[1] <= B:
                toAdd = (B - pClose[1]) / incy
                pClose[0] = pClose[0] + toAdd * incx
                pClose[1] = pClose[1] + toAdd * incy
            elif pClose[1] >= rowsB:
                toAdd = (rowsB - pClose[1]) / incy
                pClose[0] = pClose[0] + toAdd * incx
                pClose[1] = pClose[1] + toAdd * incy
            fincx = pClose[0] - pFar[0]
            fincy = pClose[1] - pFar[1]
            newEplLength = F(np.sqrt(fincx * fincx + fincy * fincy))
            if pClose[0] <= B or pClose[0] >= colsB or pClose[1] <= B or pClose[1] >= rowsB or newEplLength < F(8.0):
                return F(-1), NAN, NAN, NAN
        cpx = pFar[0]
        cpy = pFar[1]
        val_cp_m2 = interp(cur, cpx - F(2.0) * incx, cpy - F(2.0) * incy)
        val_cp_m1 = interp(cur, cpx - incx, cpy - incy)
        val_cp = interp(cur, cpx, cpy)
        val_cp_p1 = interp(cur, cpx + incx, cpy + incy)
        loopCounter = 0
        best_match_x = F(-1)
        best_match_y = F(-1)
        best_match_err = F(np.inf)            # float best_match_err = 1e50;
        second_best_match_err = F(np.inf)
        best_match_errPre = NAN; best_match_errPost = NAN; best_match_DiffErrPre = NAN; best_match_DiffErrPost = NAN
        bestWasLastLoop = False
        eeLast = F(-1)
        e1A = e1B = e2A = e2B = e3A = e3B = e4A = e4B = e5A = e5B = NAN
        loopCBest = -1
        loopCSecond = -1
        while (((incx < 0) == (cpx > pClose[0])) and ((incy < 0) == (cpy > pClose[1]))) or loopCounter == 0:
            val_cp_p2 = interp(cur, cpx + F(2) * incx, cpy + F(2) * incy)
            ee = F(0)
            if loopCounter % 2 == 0:
                e1A = val_cp_p2 - realVal_p2; ee = ee + e1A * e1A
                e2A = val_cp_p1 - realVal_p1; ee = ee + e2A * e2A
                e3A = val_cp - realVal; ee = ee + e3A * e3A
                e4A = val_cp_m1 - realVal_m1; ee = ee + e4A * e4A
                e5A = val_cp_m2 - realVal_m2; ee = ee + e5A * e5A
            else:
                e1B = val_cp_p2 - realVal_p2; ee = ee + e1B * e1B
                e2B = val_cp_p1 - realVal_p1; ee = ee + e2B * e2B
                e3B = val_cp - realVal; ee = ee + e3B * e3B
                e4B = val_cp_m1 - realVal_m1; ee = ee + e4B * e4B
                e5B = val_cp_m2 - realVal_m2; ee = ee + e5B * e5B
            if ee < best_match_err:
                second_best_match_err = best_match_err
                loopCSecond = loopCBest
                best_match_err = ee
                loopCBest = loopCounter
                best_match_errPre = eeLast
                best_match_DiffErrPre = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B
                best_match_errPost = F(-1)
                best_match_DiffErrPost = F(-1)
                best_match_x = cpx
                best_match_y = cpy
                bestWasLastLoop = True
            else:
                if bestWasLastLoop:
                    best_match_errPost = ee
                    best_match_DiffErrPost = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B
                    bestWasLastLoop = False
                if ee < second_best_match_err:
                    second_best_match_err = ee
                    loopCSecond = loopCounter
            eeLast = ee
            val_cp_m2 = val_cp_m1; val_cp_m1 = val_cp; val_cp = val_cp_p1; val_cp_p1 = val_cp_p2
            cpx = cpx + incx
            cpy = cpy + incy
            loopCounter += 1
        if best_match_err > F(4.0) * MAX_ERROR_STEREO:
            return F(-3), NAN, NAN, NAN
        if abs(loopCBest - loopCSecond) > 1.0 and MIN_DISTANCE_ERROR_STEREO * best_match_err > second_best_match_err:
            return F(-2), NAN, NAN, NAN
        didSubpixel = False
        gradPre_pre = -(best_match_errPre - best_match_DiffErrPre)
        gradPre_this = +(best_match_err - best_match_DiffErrPre)
        gradPost_this = -(best_match_err - best_match_DiffErrPost)
        gradPost_post = +(best_match_errPost - best_match_DiffErrPost)
        interpPost = False
        interpPre = False
        if best_match_errPre < 0 or best_match_errPost < 0:
            pass
        elif (gradPre_pre < 0) ^ (gradPre_this < 0):
            if (gradPost_post < 0) ^ (gradPost_this < 0):
                pass
            else:
                interpPre = True
        elif (gradPost_post < 0) ^ (gradPost_this < 0):
            interpPost = True
        if interpPre:
            d = gradPre_this / (gradPre_this - gradPre_pre)
            best_match_x = best_match_x - d * incx
            best_match_y = best_match_y - d * incy
            best_match_err = best_match_err - F(2) * d * gradPre_this - (gradPre_pre - gradPre_this) * d * d
            didSubpixel = True
        elif interpPost:
            d = gradPost_this / (gradPost_this - gradPost_post)
            best_match_x = best_match_x + d * incx
            best_match_y = best_match_y + d * incy
            best_match_err = best_match_err + F(2) * d * gradPost_this + (gradPost_post - gradPost_this) * d * d
            didSubpixel = True
        sampleDist = GRADIENT_SAMPLE_DIST * rescaleFactor
        gradAlongLine = F(0)
        tmp = realVal_p2 - realVal_p1; gradAlongLine = gradAlongLine + tmp * tmp
        tmp = realVal_p1 - realVal; gradAlongLine = gradAlongLine + tmp * tmp
        tmp = realVal - realVal_m1; gradAlongLine = gradAlongLine + tmp * tmp
        tmp = realVal_m1 - realVal_m2; gradAlongLine = gradAlongLine + tmp * tmp
        gradAlongLine = gradAlongLine / (sampleDist * sampleDist)
        if best_match_err > MAX_ERROR_STEREO + F(np.sqrt(gradAlongLine)) * F(20):
            return F(-3), NAN, NAN, NAN
        tt = self.ThisWrtOther_t
        Rr = self.ThisWrtOther_r
        if incx * incx > incy * incy:
            oldX = self.FX_INV * best_match_x + self.CX_INV
            nominator = oldX * tt[2] - tt[0]
            dot0 = dot3_fixed(KinvP, Rr[0])
            dot2 = dot3_fixed(KinvP, Rr[2])
            idnew_best_match = (dot0 - oldX * dot2) / nominator
            alpha = incx * self.FX_INV * (dot0 * tt[2] - dot2 * tt[0]) / (nominator * nominator)
        else:
            oldY = self.FY_INV * best_match_y + self.CY_INV
            nominator = oldY * tt[2] - tt[1]
            dot1 = dot3_fixed(KinvP, Rr[1])
            dot2 = dot3_fixed(KinvP, Rr[2])
            idnew_best_match = (dot1 - oldY * dot2) / nominator
            alpha = incy * self.FX_INV * (dot1 * tt[2] - dot2 * tt[1]) / (nominator * nominator)     # ORIG_FX_INV in the y branch: sic (:851)
        if idnew_best_match < 0:
            return F(-2), NAN, NAN, NAN
        photoDispError = F(4.0) * F(CAMERA_PIXEL_NOISE) / (gradAlongLine + DIVISION_EPS)
        trackingErrorFac = F(0.25) * F(1.0)
        g0 = interp(self.kf_gradx, u, v)
        g1 = interp(self.kf_grady, u, v)
        geoDispError = (g0 * epxn + g1 * epyn) + DIVISION_EPS
        geoDispError = trackingErrorFac * trackingErrorFac * (g0 * g0 + g1 * g1) / (geoDispError * geoDispError)
        result_var = alpha * alpha * ((F(0.05) if didSubpixel else F(0.5)) * sampleDist * sampleDist + geoDispError + photoDispError)
        return best_match_err, idnew_best_match, result_var, eplLength

    def _stereo(self, *a):
        r = self.do_line_stereo(*a)
        e = float(r[0])
        cls = int(e) if e in (-1.0, -2.0, -3.0, -4.0) else 0
        self.stereo_returns[cls] = self.stereo_returns.get(cls, 0) + 1
        return r

    # ---- DepthPropagation.cpp:267-308
    def observe_depth_create(self, x, y):
        st = self.st
        ep = self.make_and_check_epl(x, y)
        if ep is None:
            return -1
        error, result_idepth, result_var, _ = self._stereo(F(x), F(y), ep[0], ep[1], F(0.0), F(1.0), F(1.0) / MIN_DEPTH)
        if error == -3 or error == -2:
            st["blacklisted"][y, x] -= 1
        if error < 0 or result_var > MAX_VAR:
            return -2
        st["invDepth"][y, x] = unzero(result_idepth)
        st["variance"][y, x] = result_var
        st["invDepthSmoothed"][y, x] = -1
        st["varianceSmoothed"][y, x] = -1
        st["validity"][y, x] = VALIDITY_COUNTER_INITIAL_OBSERVE
        st["valid"][y, x] = 1
        st["blacklisted"][y, x] = 0
        return 1

    # ---- DepthPropagation.cpp:888-999
    def observe_depth_update(self, x, y):
        st = self.st
        ep = self.make_and_check_epl(x, y)
        if ep is None:
            return -5
        idS = F(st["invDepthSmoothed"][y, x])
        varS = F(st["varianceSmoothed"][y, x])
        sv = F(np.sqrt(varS))
        min_idepth = idS - sv * STEREO_EPL_VAR_FAC
        max_idepth = idS + sv * STEREO_EPL_VAR_FAC
        if min_idepth < 0:
            min_idepth = F(0)
        if max_idepth > F(1) / MIN_DEPTH:
            max_idepth = F(1) / MIN_DEPTH
        error, result_idepth, result_var, _ = self._stereo(F(x), F(y), ep[0], ep[1], min_idepth, idS, max_idepth)
        diff = result_idepth - idS
        if error == -1:
            return -1
        elif error == -2:
            vc = F(int(st["validity"][y, x])) - VALIDITY_COUNTER_DEC       # int -= float: through float, truncated back
            vc = c_int(vc)
            if vc < 0:
                vc = 0
            st["validity"][y, x] = vc
            var = F(st["variance"][y, x]) * FAIL_VAR_INC_FAC
            st["variance"][y, x] = var
            if var > MAX_VAR:
                st["valid"][y, x] = 0
                st["blacklisted"][y, x] -= 1
            return -2
        elif error == -3:
            return -3
        elif error == -4:
            return -4
        elif DIFF_FAC_OBSERVE * diff * diff > result_var + varS:
            var = F(st["variance"][y, x]) * FAIL_VAR_INC_FAC
            st["variance"][y, x] = var
            if var > MAX_VAR:
                st["valid"][y, x] = 0
            return -6
        else:
            variance = F(st["variance"][y, x])
            id_var = variance * SUCC_VAR_INC_FAC
            w = result_var / (result_var + id_var)
            new_idepth = (F(1) - w) * result_idepth + w * F(st["invDepth"][y, x])
            st["invDepth"][y, x] = unzero(new_idepth)
            id_var = id_var * w
            if id_var < variance:
                st["variance"][y, x] = id_var
            vc = c_int(F(int(st["validity"][y, x])) + VALIDITY_COUNTER_INC)
            absGrad = F(self.kf_maxgrad[y, x])
            lim = VALIDITY_COUNTER_MAX + absGrad * VALIDITY_COUNTER_MAX_VARIABLE / F(255.0)
            if F(vc) > lim:
                vc = c_int(lim)
            st["validity"][y, x] = vc
            return 1

    # ---- DepthPropagation.cpp:191-263 (one band: the bands only split independent pixels, :1932-1958)
    def observe_depth_row(self, ymin, ymax):
        st = self.st
        for y in range(ymin, ymax):
            for x in range(3, self.W - 3):
                has = bool(st["valid"][y, x])
                mg = F(self.kf_maxgrad[y, x])
                if has and mg < MIN_ABS_GRAD_DECREASE:
                    st["valid"][y, x] = 0
                    continue
                if mg < MIN_ABS_GRAD_CREATE or int(st["blacklisted"][y, x]) < MIN_BLACKLIST:
                    continue
                if not has:
                    self.observe_depth_create(x, y)
                else:
                    r = self.observe_depth_update(x, y)
                    self.update_returns[r] = self.update_returns.get(r, 0) + 1

    # ---- DepthPropagation.cpp:1003-1157. new_kf_image / new_kf_maxgrad: the NEW keyframe's; mats: new_keyframe->calculateSE3poseOtherWrtThis(keyFrame)
    def propagate_depth(self, new_kf_image, new_kf_maxgrad, mats):
        W, H = self.W, self.H
        cur = self.st
        other = {k: np.array(v, copy=True) for k, v in cur.items()}     # the other buffer: whatever it held, then wiped
        other["valid"][:] = 0
        other["blacklisted"][:] = 0
        R = [[F(mats["ThisWrtOther"][r, c]) for c in range(3)] for r in range(3)]
        t = [F(mats["ThisWrtOther"][r, 3]) for r in range(3)]
        src_img = self.kf_image
        for y in range(H):
            for x in range(W):
                if not cur["valid"][y, x]:
                    continue
                idS = F(cur["invDepthSmoothed"][y, x])
                p = [F(x) * self.FX_INV + self.CX_INV, F(y) * self.FY_INV + self.CY_INV, F(1.0)]
                Rp = matvec3(R, p)                # dynamic-size product (MatrixXf): column by column, the same left-to-right sums
                pn = [Rp[i] / idS + t[i] for i in range(3)]
                new_idepth = F(1.0) / pn[2]
                u_new = pn[0] * new_idepth * self.FX + self.CX
                v_new = pn[1] * new_idepth * self.FY + self.CY
                if not (u_new > F(2.1) and v_new > F(2.1) and u_new < F(W) - F(3.1) and v_new < F(H) - F(3.1)):
                    continue
                newX = c_int(u_new + F(0.5)); newY = c_int(v_new + F(0.5))
                destAbsGrad = F(new_kf_maxgrad[y, x])               # sic: the NEW keyframe's plane at the SOURCE pixel
                sourceColor = F(src_img[y, x])
                destColor = interp(new_kf_image, u_new, v_new)
                residual = destColor - sourceColor
                if residual * residual / (MAX_DIFF_CONSTANT + MAX_DIFF_GRAD_MULT * destAbsGrad * destAbsGrad) > F(1.0) or destAbsGrad < MIN_ABS_GRAD_DECREASE:
                    continue
                r4 = new_idepth / idS
                r4 = r4 * r4
                r4 = r4 * r4
                new_var = r4 * F(cur["invDepth"][y, x])             # sic: source->invDepth, not its variance (:1086)
                if other["valid"][newY, newX]:
                    diff = F(other["invDepth"][newY, newX]) - new_idepth
                    if DIFF_FAC_PROP_MERGE * diff * diff > new_var + F(other["variance"][newY, newX]):
                        if new_idepth < F(other["invDepth"][newY, newX]):
                            continue
                        else:
                            other["valid"][newY, newX] = 0
                if not other["valid"][newY, newX]:
                    other["invDepth"][newY, newX] = new_idepth
                    other["variance"][newY, newX] = new_var
                    other["varianceSmoothed"][newY, newX] = -1
                    other["invDepthSmoothed"][newY, newX] = -1
                    other["validity"][newY, newX] = cur["validity"][y, x]
                    other["valid"][newY, newX] = 1
                    other["blacklisted"][newY, newX] = 0
                else:
                    tv = F(other["variance"][newY, newX])
                    w = new_var / (tv + new_var)
                    merged = w * F(other["invDepth"][newY, newX]) + (F(1.0) - w) * new_idepth
                    mv = int(cur["validity"][y, x]) + int(other["validity"][newY, newX])
                    if F(mv) > VALIDITY_COUNTER_MAX + VALIDITY_COUNTER_MAX_VARIABLE:      # int compared with a float sum
                        mv = c_int(VALIDITY_COUNTER_MAX + VALIDITY_COUNTER_MAX_VARIABLE)
                    other["invDepth"][newY, newX] = merged
                    other["variance"][newY, newX] = F(1.0) / (F(1.0) / tv + F(1.0) / new_var)
                    other["validity"][newY, newX] = mv
                    other["valid"][newY, newX] = 1
                    other["blacklisted"][newY, newX] = 0
                    other["invDepthSmoothed"][newY, newX] = F(-1.0)
                    other["varianceSmoothed"][newY, newX] = F(-1.0)
        self.st = other          # std::swap(currentDepthHypothesis, otherDepthHypothesis)
        return other

// ELLC hot-path ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A dependency-free CPU restatement of the reference algorithm
// (IITD-COMPUTER-VISION-GROUP/Egomotion_with_Local_Loop_Closures, files cited
// per function as  <file>:<line>  relative to the reference's src/ directory).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// link or call anything in oracle/.  The product path (the HIP library in
// egomotion_with_local_loop_closures_amd/csrc) never includes this header.
//
// PARITY STATUS: "parity unpinned" by the reference.  The reference ships no
// tests, no golden vectors and no data, and cannot be compiled in the build
// container (needs OpenCV 3.0.0, Eigen 3.2.5 + unsupported/MatrixFunctions,
// Boost.Thread 1.59 — none installed, no network).  The oracle is pinned by
//   * analytic known-answer tests (tests/test_oracle_*.py),
//   * scipy expm/logm goldens for the se(3) exp/log (tests/golden/),
//   * committed golden vectors generated from this restatement.
// Third-party arithmetic restated from the libraries' published algorithms:
//   OpenCV 3.0.0  cv::pyrDown (8u, 5x5 [1 4 6 4 1], BORDER_REFLECT_101,
//                 (sum+128)>>8), cv::Mat::inv(DECOMP_LU) for 6x6 f32 (LU with
//                 partial pivoting, |pivot| < FLT_EPSILON => singular => zero
//                 matrix), 3x3 f32 closed-form inverse with double det,
//                 cv::gemm f32 with double accumulation.
//   Eigen 3.2.5   MatrixBase::exp()/log() on 4x4 f32: restated as the closed
//                 form (Rodrigues + V matrix) evaluated in double and rounded
//                 to f32 (differs from Eigen's f32 Pade/Schur at ~1e-7 rel.).
#pragma once
#include <cstdint>
#include <vector>
#include <cstddef>

namespace ellc_oracle {

constexpr int kMaxLevels = 8;

// ExternVariable.h:232  UNZERO(val): clamp |val| >= 1e-10 keeping sign; the
// literals are double so the selected value is a double rounded to f32 on store.
inline float unzero(float v) {
  double r = (v < 0) ? ((v > -1e-10) ? -1e-10 : (double)v) : ((v < 1e-10) ? 1e-10 : (double)v);
  return (float)r;
}

template <class T>
struct Plane {
  int w = 0, h = 0;
  std::vector<T> d;
  Plane() {}
  Plane(int w_, int h_, T v = T()) : w(w_), h(h_), d((size_t)w_ * h_, v) {}
  T& at(int y, int x) { return d[(size_t)y * w + x]; }
  const T& at(int y, int x) const { return d[(size_t)y * w + x]; }
  T* row(int y) { return d.data() + (size_t)y * w; }
  const T* row(int y) const { return d.data() + (size_t)y * w; }
};
typedef Plane<uint8_t> PlaneU8;
typedef Plane<float> PlaneF;

// ---- run-time replacement for the compile-time constants of ExternVariable.h
struct Config {
  int width = 640, height = 480, levels = 4;  // ExternVariable.h:40,50-51
  float fx = 547.2f, fy = 547.2f, cx = 320.f, cy = 240.f;  // ExternVariable.h:53-59
  int max_iter[kMaxLevels] = {4, 7, 9, 12, 12, 12, 12, 12};  // main.cpp:34 (index = level)
  int early_exit = 1;       // ImageFunc.cpp:251-252
  int num_pose_threads = 3; // ExternVariable.h:224 (row bands of the FCA sum)
};

struct Intrin { float fx, fy, cx, cy; };
// UserDefinedFunc.cpp:34-50  (division by pow(2,l) in double, stored f32)
Intrin get_intrinsic(const Config& c, int level);

// ---------------------------------------------------------------- se(3) / SE(3)
// 4x4 row-major f32 matrices, pose = [wx wy wz vx vy vz]  (PixelWisePyramid.cpp:153)
void se3_exp(const float pose[6], float T[16]);            // Eigen .exp()  PixelWisePyramid.cpp:157
void se3_log(const float T[16], float pose[6]);            // Eigen .log() + vee  Frame.cpp:521-528
void se3_exp_d(const double pose[6], double T[16]);
void se3_log_d(const double T[16], double pose[6]);
void concatenate_relative_pose(const float a[6], const float b[6], float out[6]);  // Frame.cpp:503-530
void concatenate_origin_pose(const float a[6], const float b[6], float out[6]);    // Frame.cpp:534-562
void inv_lie_pose(const float a[6], float out[6]);                                 // Frame.cpp:592-615

// cv::Mat::inv(DECOMP_LU) for an n x n f32 matrix (n = 6 on the path). Returns 0 when singular
// (then out is all zeros)  PixelWisePyramid.cpp:451
int lu_inverse_f32(const float* A, int n, float* out);

// ---------------------------------------------------------------- image side
void pyr_down_u8(const PlaneU8& src, PlaneU8& dst);                     // cv::pyrDown  Frame.cpp:175-179
void calculate_gradient(const PlaneU8& img, int rows, int cols, PlaneF& gx, PlaneF& gy);  // Frame.cpp:185-285
void build_max_gradients(const PlaneF& gx, const PlaneF& gy, PlaneF& out, int* n_substantial);  // Frame.cpp:618-674
// Frame.h:181-279 (u8 tap; returns -1 iff all 4 taps OOB and check==1)
float tap_u8(const PlaneU8& img, int curRows, int curCols, float x1, float y1, int check);
// Frame.h:283-394 (f32 gradient tap; never signals OOB)
float tap_f32(const PlaneF& img, int curRows, int curCols, float x1, float y1);

struct Frame {
  Config cfg;
  int frameId = 0;
  int width = 0, height = 0;
  std::vector<PlaneU8> image_pyramid;   // stored sizes follow pyrDown's ceil rule (Q13)
  std::vector<PlaneF> depth_pyramid;    // sizes (h>>l, w>>l)  Frame.cpp:109-112
  std::vector<PlaneF> weight_pyramid;   // Frame.cpp:114-117
  int numWeightsAdded[kMaxLevels] = {0};
  PlaneF gradientx, gradienty;          // of image_pyramid[pyrLevel], size currentRows x currentCols
  PlaneF maxAbsGradient;                // level 0
  PlaneU8 mask;
  int no_points_substantial_grad = 0;
  int currentRows = 0, currentCols = 0, pyrLevel = 0, no_nonZeroDepthPts = 0;
  float poseWrtOrigin[6] = {0}, poseWrtWorld[6] = {0};
  float rescaleFactor = 1.0f;
  // Frame.cpp:376-413
  float SE3poseOtherWrtThis[16], SE3poseThisWrtOther[16];
  float K_SE3poseThisWrtOther_r[9], K_SE3poseThisWrtOther_t[3];
  float K_SE3poseOtherWrtThis_r[9], K_SE3poseOtherWrtThis_t[3];

  void init(const Config& c, const uint8_t* gray, int id);  // Frame.cpp:78-123 (after decode/undistort)
  void constructImagePyramids();                            // Frame.cpp:170-182
  void calculateGradient();                                 // Frame.cpp:185-285
  void calculateNonZeroDepthPts();                          // Frame.cpp:295-301
  void updationOnPyrChange(int level, bool isPrevious = true);  // Frame.cpp:316-327
  void buildMaxGradients();                                 // Frame.cpp:618-674
  void finaliseWeights();                                   // Frame.cpp:678-695
  void calculateSE3poseOtherWrtThis(const Frame& other);    // Frame.cpp:376-413
  float getInterpolatedElement(float x, float y, int check = 0) const {
    return tap_u8(image_pyramid[pyrLevel], currentRows, currentCols, x, y, check);
  }
  float getInterpolatedGradX(float x, float y) const { return tap_f32(gradientx, currentRows, currentCols, x, y); }
  float getInterpolatedGradY(float x, float y) const { return tap_f32(gradienty, currentRows, currentCols, x, y); }
};

// EigenInitialization.cpp:20-34  K, Kinv (cv 3x3 f32 inverse with double determinant)
struct KMats { float K[9]; float Kinv[9]; float fx_inv, fy_inv, cx_inv, cy_inv; };
KMats make_kmats(const Config& c);

// ---------------------------------------------------------------- depth variance source for the GN weights
struct DepthPyr {  // depthMap::deptharrptr / depthvararrptr  DepthPropagation.h:65-78
  std::vector<std::vector<float>> deptharr, depthvararr;
};

// ---------------------------------------------------------------- Gauss-Newton
enum SumMode { SUM_F32_BANDS = 0, SUM_F64 = 1 };

struct GNDebugPlanes {  // PixelWisePyramid.cpp:26-37 display_* planes + per-pixel J
  PlaneF residual, weight, warpedX, warpedY, warped;
  std::vector<PlaneF> J;  // 6 planes
};

struct PixelWisePyramid {
  Frame* prev_frame;     // keyframe (template)
  Frame* current_frame;
  const DepthPyr* depthMap;
  float* pose;           // PixelWisePyramid.h: caller assigns .pose (ImageFunc.cpp:183)
  int pyrlevel, nRows, nCols;
  float hessian[36], sd_param[6], hessianInv[36], deltapose[6];
  double hessian_d[36], sd_param_d[6];
  float weightedPose = 0;
  SumMode sum_mode = SUM_F32_BANDS;
  int n_threads = 3;         // NUM_POSE_THREADS
  int thread_mode = 0;       // 0: bands run one after the other on the caller's thread (tests); 1: std::threads created
                             // and joined per iteration as the reference does (CPU baseline); 2: persistent worker pool
                             // (CPU baseline variant). Results identical in all three.
  PlaneF display_weightimg, display_iterationres;
  PlaneF savedWarpedPointsX, savedWarpedPointsY;
  std::vector<float> steepestDescent, weightedSteepestDescent;  // 6 x N row-major (ICA)
  GNDebugPlanes* dbg = nullptr;

  PixelWisePyramid(Frame* prev, Frame* cur, float* pose, const DepthPyr* dm);
  void calculatePixelWise(int ymin, int ymax, float H[36], float b[6], double Hd[36], double bd[6]);  // :58-413
  void calculatePixelWiseParallel();                        // :416-455
  void updatePose();                                        // :460-491
  void saveWeights(bool useAverageWeights);                 // :500-552
  void precomputePixelWiseInvCompositional(int ymin, int ymax);                  // :561-680
  void iteratePixelWiseInvCompositional(int ymin, int ymax, float b[6], double bd[6]);  // :687-913
  void calculatePixelWiseParallelInvCompositional(int iter);                     // :917-974
};

struct AlignResult {
  float pose[6];
  int iters[kMaxLevels];
  float last_weighted;
};
// ImageFunc.cpp:49-315  (initial pose from tminus1/prev world poses unless init_pose given)
AlignResult GetImagePoseEstimate(Frame* prev_frame, Frame* current_frame, const DepthPyr* dm,
                                 Frame* tminus1, const float* init_rel_pose, bool fromLoopClosure,
                                 bool save_weights, SumMode mode, int thread_mode, int n_threads);

// ---------------------------------------------------------------- depth map
struct Hyp {  // DepthHypothesis.h:14-40 (live fields only)
  float invDepth = 0, invDepthSmoothed = 0, variance = 0, varianceSmoothed = 0;
  int validity_counter = 0, blacklisted = 0;
  uint8_t isValid = 0;
};

struct DepthMap {
  Config cfg;
  KMats km;
  int W, H;
  std::vector<Hyp> current, other;          // DepthPropagation.h:43-45
  std::vector<int> validityIntegralBuffer;  // DepthPropagation.h:47
  DepthPyr pyr;                             // deptharrpyr*/depthvararrpyr*
  Frame* keyFrame = nullptr;
  Frame* currentFrame = nullptr;
  float depthScale = 1.f;
  float global_depth_scale = 1.f;           // util::GLOABL_DEPTH_SCALE made per-object

  void init(const Config& c);
  void propagateDepth(Frame* new_keyframe);                 // :1003-1157
  void observeDepthRow(int ymin, int ymax);                 // :191-263
  void observeDepthRowParallel();                           // :1932-1958
  int observeDepthCreate(int x, int y, int idx);            // :267-308
  int observeDepthUpdate(int x, int y, int idx);            // :888-999
  bool makeAndCheckEPL(int x, int y, float* pepx, float* pepy);  // :311-384
  float doLineStereo(float u, float v, float epxn, float epyn, float min_idepth, float prior_idepth,
                     float max_idepth, float& result_idepth, float& result_var, float& result_eplLength);  // :397-885
  void buildValIntegralBuffer();                            // :1403-1432
  void fillDepthHoles();                                    // :1317-1400
  void regularizeDepthMap(bool removeOcclusions);           // :1436-1543
  void doRegularization(bool removeOcclusions = false) { fillDepthHoles(); regularizeDepthMap(removeOcclusions); }  // :1627-1635
  void makeInvDepthOne();                                   // :1546-1587
  void updateDepthImage();                                  // :1254-1315 (no display)
  void buildInvVarDepth();                                  // :1637-1719
  void mapDepthArr2Mat();                                   // :1722-1746
  float calculate_no_of_Seeds() const;                      // :1804-1830
  void createKeyFrame(Frame* new_keyframe);                 // :1758-1794
  void finaliseKeyframe() { doRegularization(); updateDepthImage(); }  // :1749-1755
};

}  // namespace ellc_oracle

// Semi-dense depth map entry points (class depthMap, DepthPropagation.cpp). Placeholder bodies: the
// kernels land in the next milestone; until then every call reports ELLC_ERR_NOT_READY loudly.
#include "ellc_context.hpp"
using namespace ellc;
extern "C" {
#define NOTYET(c) return fail(c, ELLC_ERR_NOT_READY, "depth-map kernels not built in this library version")
ellc_status ellc_depth_set_state(ellc_ctx* c, const ellc_hypotheses*) { NOTYET(c); }
ellc_status ellc_depth_get_state(ellc_ctx* c, const ellc_hypotheses*) { NOTYET(c); }
ellc_status ellc_depth_set_keyframe(ellc_ctx* c, int) { NOTYET(c); }
ellc_status ellc_depth_propagate(ellc_ctx* c, int, const float*) { NOTYET(c); }
ellc_status ellc_depth_observe(ellc_ctx* c, int, const float*) { NOTYET(c); }
ellc_status ellc_depth_fill_holes(ellc_ctx* c) { NOTYET(c); }
ellc_status ellc_depth_regularize(ellc_ctx* c, int) { NOTYET(c); }
ellc_status ellc_depth_make_inv_depth_one(ellc_ctx* c, float*) { NOTYET(c); }
ellc_status ellc_depth_update_depth_image(ellc_ctx* c) { NOTYET(c); }
ellc_status ellc_depth_create_keyframe(ellc_ctx* c, int, const float*, float*) { NOTYET(c); }
ellc_status ellc_depth_seeds(ellc_ctx* c, float*) { NOTYET(c); }
}

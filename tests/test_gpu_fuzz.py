"""Randomised parity sweep: many small scenes (sizes incl. odd ones, seeds, poses far from the identity, custom
intrinsics) through the per-pixel stages that are claimed bit-exact — FCA planes, ICA planes, and the four depth stages.
Each case is tiny; the sweep exists to hit the rare branches a single scene does not (found in r01: a 1-ulp pose-algebra
difference that only showed on a 96x64 scene)."""
import numpy as np
import pytest
from egomotion_with_local_loop_closures_amd import synth
from helpers import oracle_problem, bits_equal

pytestmark = pytest.mark.gpu

CASES = [(96, 64, 3, 101), (101, 75, 3, 102), (64, 48, 3, 103), (160, 120, 4, 104), (128, 72, 3, 105), (90, 110, 3, 106),
         (200, 64, 3, 107), (64, 200, 3, 108), (96, 64, 3, 109), (96, 64, 3, 110), (117, 83, 3, 111), (240, 136, 4, 112),
         (72, 56, 3, 113), (150, 100, 3, 114), (96, 96, 3, 115), (320, 64, 3, 116)]
# ELLC_FUZZ_EXTRA=n appends n more random (size, seed) cases for a one-off wide sweep (r01: 300 extra cases, all green)
import os   # noqa: E402
_extra = int(os.environ.get("ELLC_FUZZ_EXTRA", "0"))
if _extra:
    _r = np.random.default_rng(12345)
    CASES = CASES + [(int(_r.integers(48, 260)), int(_r.integers(48, 200)), 3, 1000 + i) for i in range(_extra)]
FIELDS = ("invDepth", "invDepthSmoothed", "variance", "varianceSmoothed", "validity", "blacklisted")


def states_equal(got, ref, what):
    m = ref["valid"] != 0
    assert np.array_equal(got["valid"] != 0, m), what + ": valid mask"
    assert np.array_equal(got["blacklisted"], ref["blacklisted"]), what + ": blacklisted"
    for f in FIELDS[:5]:
        assert bits_equal(got[f][m], ref[f][m]), "%s: %s (%d px differ)" % (what, f, int((got[f][m] != ref[f][m]).sum()))


@pytest.mark.parametrize("w,h,L,seed", CASES)
def test_random_scene_per_pixel_parity(oracle, ellc, w, h, L, seed):
    rng = np.random.default_rng(seed)
    rot = float(rng.uniform(0.002, 0.03)); trans = float(rng.uniform(0.005, 0.06))
    pair = synth.make_pair(w, h, seed=seed, rot=rot, trans=trans)
    # intrinsics away from the defaults (non-square pixels, off-centre principal point)
    fx, fy, cx, cy = pair["intrinsics"]
    pair["intrinsics"] = (fx * float(rng.uniform(0.8, 1.3)), fy * float(rng.uniform(0.8, 1.3)), cx + float(rng.uniform(-4, 4)),
                          cy + float(rng.uniform(-3, 3)))
    fx, fy, cx, cy = pair["intrinsics"]
    ocfg, kf, cur, dm = oracle_problem(oracle, w, h, L, pair)
    ctx = ellc.Context(ellc.default_config(w, h, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=1))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"]); ctx.frame_upload(0, pair["cur_image"])
    # ---- Gauss-Newton planes at every level, poses up to ~0.1 rad / 0.2 units away from the identity
    for level in reversed(range(L)):   # coarse to fine: tracking leaves both frames at level 0, which the depth code relies on (Q11)
        pose = (rng.normal(size=6) * [0.03, 0.03, 0.03, 0.06, 0.06, 0.06]).astype(np.float32)
        mask = kf.depth(level) > 0
        if mask.sum() < 20:
            continue
        st = oracle.GNStepper(kf, cur, dm.depth_pyr(), level, pose, planes=True)
        st.step(0)
        pl = st.get_planes()
        got = ctx.gn_iterate(0, 0, level, pose, planes=True)
        for name in ("residual", "weight", "warpedX", "warpedY"):
            assert bits_equal(got[name][mask], pl[name][mask]), (level, name)
        for k in range(6):
            assert bits_equal(got["J"][k][mask], pl["J"][k][mask]), (level, "J%d" % k)
        st.close()
    # ---- constant-weight path: template-gradient Jacobian and residual at one level
    for l in range(L):
        wplane = rng.uniform(0.01, 0.0625, size=(h >> l, w >> l)).astype(np.float32)
        kf.set_weights(l, wplane, 1); ctx.keyframe_set_weights(0, l, wplane, 1)
    lvl = int(rng.integers(0, L))
    pose = (rng.normal(size=6) * [0.01, 0.01, 0.01, 0.03, 0.03, 0.03]).astype(np.float32)
    mask = kf.depth(lvl) > 0
    if mask.sum() >= 20:
        st = oracle.GNStepper(kf, cur, dm.depth_pyr(), lvl, pose, planes=True)
        st.step(1, 0)
        sd, _ = st.get_sd()
        pl = st.get_planes()
        got = ctx.gn_iterate(0, 0, lvl, pose, mode=1, it=0, planes=True)
        sd = sd.reshape(6, h >> lvl, w >> lvl)
        for k in range(6):
            assert bits_equal(got["J"][k][mask], sd[k][mask]), ("ica J%d" % k, lvl)
        assert bits_equal(got["residual"][mask], pl["residual"][mask]), ("ica residual", lvl)
        st.close()
        if lvl != 0:   # leave the oracle frames at level 0 for the depth code (Q11)
            st = oracle.GNStepper(kf, cur, dm.depth_pyr(), 0, pose, planes=False)
            st.close()
    # ---- depth stages on a hypothesis map of the same scene, pose = the scene's true motion plus noise
    xi = (np.asarray(pair["xi_true"]) + rng.normal(size=6) * 1e-3).astype(np.float32)
    stt = synth.make_depth_state(w, h, seed + 1, pair["kf_image"], pair["idepth_true"])
    ocur = oracle.Frame(ocfg, pair["cur_image"], 2)
    ocur.set_pose(origin=xi, world=xi)
    ctx.keyframe_from_frame(1, 0)

    def fresh():
        d = oracle.DepthMap(ocfg)
        d.set_keyframe(kf); d.set_current(ocur); d.set_state(stt)
        ctx.depth_set_keyframe(0); ctx.depth_set_state(stt)
        return d
    d = fresh(); d.regularize(True); ctx.depth_regularize(True); states_equal(ctx.depth_get_state(), d.get_state(), "regularize(occl)")
    d.observe(); ctx.depth_observe(0, xi); states_equal(ctx.depth_get_state(), d.get_state(), "observe")
    d.fill_holes(); ctx.depth_fill_holes(); states_equal(ctx.depth_get_state(), d.get_state(), "fill_holes")
    d = fresh(); d.regularize(False); ctx.depth_regularize(False)
    nk = oracle.Frame(ocfg, pair["cur_image"], 3)
    nk.set_pose(origin=xi)
    d.propagate(nk); ctx.depth_propagate(1, xi); states_equal(ctx.depth_get_state(), d.get_state(), "propagate")
    ctx.close()


def _stable_case(oracle, w, h, L, seed, mi, ica):
    """First random scene (and, for the constant-weight path, weight planes) on which the ORACLE is stable: its result moves
    by <= 1e-6 when its three row-band f32 sums are replaced by f64 sums, and it converges to the scene's motion."""
    for attempt in range(16):
        sd = seed + 1000 * attempt + (500 if ica else 0)
        rng = np.random.default_rng(sd)
        pair = synth.make_pair(w, h, seed=sd, rot=float(rng.uniform(0.002, 0.012)), trans=float(rng.uniform(0.005, 0.03)))
        ocfg, kf, cur, dm = oracle_problem(oracle, w, h, L, pair, early_exit=0, max_iter=mi)
        init = (rng.normal(size=6) * [0.002, 0.002, 0.002, 0.005, 0.005, 0.005]).astype(np.float32)
        planes = None
        if ica:
            planes = [rng.uniform(0.01, 0.0625, size=(h >> l, w >> l)).astype(np.float32) for l in range(L)]
            for l in range(L):
                kf.set_weights(l, planes[l], 1)
        pr, itr, _ = oracle.align(kf, cur, dm.depth_pyr(), init_pose=init, loop_closure=ica)
        pr64, _, _ = oracle.align(kf, cur, dm.depth_pyr(), init_pose=init, loop_closure=ica, sum_mode=1)
        if np.linalg.norm(pr64 - pr) <= 1e-6 and np.linalg.norm(pr - np.asarray(pair["xi_true"], np.float32)) < 0.02:
            return pair, init, planes, pr, itr
    pytest.fail("no well-conditioned %s scene in 16 attempts" % ("ICA" if ica else "FCA"))


ALIGN_CASES = [(96, 64, 3, 201), (101, 75, 3, 202), (160, 120, 4, 203), (128, 72, 3, 204), (240, 136, 4, 205), (64, 48, 3, 206)]
if _extra:   # the wide sweep: one further full-alignment case per ten per-pixel cases
    _r2 = np.random.default_rng(54321)
    ALIGN_CASES = ALIGN_CASES + [(int(_r2.integers(64, 260)), int(_r2.integers(48, 200)), 3, 2000 + i) for i in range(_extra // 10)]


@pytest.mark.parametrize("w,h,L,seed", ALIGN_CASES)
@pytest.mark.parametrize("ica", [False, True], ids=["fca", "ica"])
@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_random_scene_full_alignment(oracle, ellc, w, h, L, seed, ica, arith):
    """Full fixed-schedule alignments of random scenes, FCA and ICA, both arithmetic modes: final pose within 1e-5 of the
    oracle, no exceptions.
    A tiny random scene can be ill-conditioned — the iteration does not contract and amplifies the rounding of the sums, the
    one thing that legitimately differs between product and oracle. Such a case is not excused with a wider bar: it is
    REGENERATED (_stable_case) until the oracle itself is stable; the first stable case is then held to 1e-5."""
    mi = (4, 7, 9, 12)[:L]
    pair, init, planes, pr, itr = _stable_case(oracle, w, h, L, seed, mi, ica)
    fx, fy, cx, cy = pair["intrinsics"]
    ctx = ellc.Context(ellc.default_config(w, h, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_iter=mi,
                                           arith=ellc.ARITH_FAST if arith == "fast" else ellc.ARITH_EXACT))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"]); ctx.frame_upload(0, pair["cur_image"])
    if ica:
        for l in range(L):
            ctx.keyframe_set_weights(0, l, planes[l], 1)
    pg, itg, _ = ctx.align([0], [0], init_pose=init[None], mode=1 if ica else 0)
    assert list(itg[0]) == list(itr) == list(mi)
    assert np.linalg.norm(pg[0] - pr) <= 1e-5, (pg[0], pr)
    ctx.close()


DENSE_CASES = [(320, 200, 3, 301), (322, 202, 3, 302), (404, 300, 4, 303), (333, 250, 3, 304), (480, 270, 4, 305)]   # (large enough that the 3-pixel border leaves nine tenths)


@pytest.mark.parametrize("w,h,L,seed", DENSE_CASES)
@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_random_dense_map_list_free_path(oracle, ellc, w, h, L, seed, arith):
    """The list-free schedules on random DENSE maps of odd geometry (widths that are and are not multiples of four, levels too small
    for the four-pixel kernel, scattered holes): gn_fca_dense / gn_fca_dense4 (tolerance mode), gn_fca_dense_x (exact mode, r06).
    Final pose within 1e-5 of the oracle's; the same planes through the compact lists (ellc_ctx_set_dense_maps(1)) within 2e-6 of
    the list-free result (the sums' order is all that differs), iteration counts equal."""
    mi = (4, 7, 9, 12)[:L]
    rng = np.random.default_rng(seed)
    pair = dict(synth.make_pair(w, h, seed=seed, dense=True, rot=float(rng.uniform(0.002, 0.01)), trans=float(rng.uniform(0.005, 0.025))))
    d0 = pair["depth0"].copy()
    d0[rng.random(d0.shape) < 0.01] = 0.0
    pair["depth0"] = d0
    assert (d0 > 0).mean() > 0.91   # (the dense hint: at least nine tenths)
    _, kf, cur, dm = oracle_problem(oracle, w, h, L, pair, early_exit=0, max_iter=mi)
    pr, itr, _ = oracle.align(kf, cur, dm.depth_pyr())
    fx, fy, cx, cy = pair["intrinsics"]
    res = []
    for pin in (0, 1):
        ctx = ellc.Context(ellc.default_config(w, h, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_iter=mi,
                                               arith=ellc.ARITH_FAST if arith == "fast" else ellc.ARITH_EXACT))
        ctx.set_dense_maps(pin)
        ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"]); ctx.frame_upload(0, pair["cur_image"])
        pg, itg, _ = ctx.align([0], [0])
        assert list(itg[0]) == list(itr) == list(mi)
        assert np.linalg.norm(pg[0] - pr) <= 1e-5, (pin, pg[0], pr)
        res.append(pg[0].copy())
        ctx.close()
    assert np.linalg.norm(res[0] - res[1]) <= 2e-6, (res[0], res[1])

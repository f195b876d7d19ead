"""Parity at BASELINE.json's full sizes (C1, C2, C4) plus size-independent properties of the batched path."""
import numpy as np
import pytest
from egomotion_with_local_loop_closures_amd import synth
from helpers import oracle_problem, gpu_problem, bits_equal

pytestmark = pytest.mark.gpu


def test_c1_640x480_single_alignment_and_planes(oracle, ellc):
    """configs[1]: one keyframe vs one frame, 640x480, 4 levels — per-pixel planes bit-exact at level 0, pose <= 1e-5."""
    W, H, L = 640, 480, 4
    pair = synth.make_pair(W, H, seed=0x5EED)
    _, kf, cur, dm = oracle_problem(oracle, W, H, L, pair)
    ctx = gpu_problem(ellc, W, H, L, [pair])
    pose = np.array([0.002, -0.001, 0.0015, 0.005, -0.004, 0.003], np.float32)
    st = oracle.GNStepper(kf, cur, dm.depth_pyr(), 0, pose, planes=True)
    ref = st.step(0)
    pl = st.get_planes()
    got = ctx.gn_iterate(0, 0, 0, pose, planes=True)
    mask = kf.depth(0) > 0
    assert 0.15 < mask.mean() < 0.4                      # semi-dense
    for name in ("residual", "weight", "warpedX", "warpedY"):
        assert bits_equal(got[name][mask], pl[name][mask]), name
    for k in range(6):
        assert bits_equal(got["J"][k][mask], pl["J"][k][mask])
    Hs = 0.5 * (ref["Hd"] + ref["Hd"].T)
    assert np.allclose(got["H"], Hs, rtol=2e-6, atol=2e-7 * np.abs(Hs).max())   # small off-diagonals are cancelling sums
    st.close()
    p_ref, it_ref, _ = oracle.align(kf, cur, dm.depth_pyr())
    p, it, _ = ctx.align([0], [0])
    assert list(it[0]) == list(it_ref)
    err = np.linalg.norm(p[0] - p_ref)
    print("C1 pose error vs oracle: %.2e" % err)
    assert err <= 1e-5
    ctx.close()


@pytest.mark.parametrize("concurrent", [1, 3])
def test_c2_batch32_matches_singles_and_oracle(oracle, ellc, concurrent):
    """configs[2]: 32 alignments in one launch sequence, on the full-round grid (one batch at a time) and on the grid
    bench.py runs (cfg.concurrent_batches = 3, with two more batches in flight on the same data). Properties: each result
    equals the single-alignment run (independence), is invariant under permutation of the batch, does not depend on what
    else is in flight, and (spot check) matches the oracle."""
    W, H, L, B = 640, 480, 4, 32
    pairs = [synth.make_pair(W, H, seed=900 + i) for i in range(4)]
    ctx = gpu_problem(ellc, W, H, L, [pairs[b % 4] for b in range(B)], concurrent_batches=concurrent)
    slots = np.arange(B)
    p_all, it_all, _ = ctx.align(slots, slots)
    assert it_all.sum() == B * 32
    if concurrent > 1:   # three batches over disjoint halves / quarters of the slots, concurrently
        parts = [slots[:16], slots[16:24], slots[24:]]
        alone = [ctx.align(q, q)[0] for q in parts]
        for q in parts:
            ctx.align_enqueue(q, q)
        for q, ref in zip(parts, alone):
            assert np.array_equal(ctx.align_fetch(len(q))[0], ref)
    perm = np.random.default_rng(0).permutation(B)
    p_perm, _, _ = ctx.align(slots[perm], slots[perm])
    assert np.array_equal(p_perm, p_all[perm])           # block decomposition does not depend on the position in the batch
    for b in (0, 1, 2, 3, 17, 31):
        p1, _, _ = ctx.align([b], [b])
        assert np.linalg.norm(p1[0] - p_all[b]) < 2e-6
    for b in range(4):                                   # slots b, b+4, ... hold the same scene
        assert np.array_equal(p_all[b], p_all[b + 4])
        _, kf, cur, dm = oracle_problem(oracle, W, H, L, pairs[b])
        p_ref, _, _ = oracle.align(kf, cur, dm.depth_pyr())
        assert np.linalg.norm(p_all[b] - p_ref) <= 1e-5
    ctx.close()


def test_c4_1280x960_dense_five_levels(oracle, ellc):
    """configs[4] shape: 1280x960, 5 levels {4,7,9,12,12}, dense (all-pixel) residuals."""
    W, H, L = 1280, 960, 5
    mi = (4, 7, 9, 12, 12)
    pair = synth.make_pair(W, H, seed=77, dense=True)
    _, kf, cur, dm = oracle_problem(oracle, W, H, L, pair, max_iter=mi)
    ctx = gpu_problem(ellc, W, H, L, [pair], max_iter=mi)
    assert (kf.depth(0) > 0).mean() > 0.98
    p_ref, it_ref, _ = oracle.align(kf, cur, dm.depth_pyr())
    p, it, _ = ctx.align([0], [0])
    assert list(it[0]) == list(it_ref) == list(mi)
    err = np.linalg.norm(p[0] - p_ref)
    print("C4 pose error vs oracle: %.2e" % err)
    assert err <= 1e-5
    ctx.close()


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_c4_batch16_1280x960_dense(oracle, ellc, arith):
    """configs[4] at its per-GPU batch: 16 alignments of 1280x960, 5 levels {4,7,9,12,12}, dense residuals, two distinct
    scenes alternating over the batch, in both arithmetic modes. Every alignment within 1e-5 of the oracle's pose for its
    scene; same scene => same bits wherever it sits in the batch; a permuted batch gives the permuted results; a batch in
    flight beside two others gives the same bits as alone."""
    W, H, L, B = 1280, 960, 5, 16
    mi = (4, 7, 9, 12, 12)
    pairs = [synth.make_pair(W, H, seed=770 + i, dense=True) for i in range(2)]
    refs = []
    for p in pairs:
        _, kf, cur, dm = oracle_problem(oracle, W, H, L, p, max_iter=mi)
        refs.append(oracle.align(kf, cur, dm.depth_pyr())[0])
    ctx = gpu_problem(ellc, W, H, L, [pairs[b % 2] for b in range(B)], max_iter=mi, concurrent_batches=3,
                      arith=ellc.ARITH_FAST if arith == "fast" else ellc.ARITH_EXACT)
    slots = np.arange(B, dtype=np.int32)
    p_all, it_all, _ = ctx.align(slots, slots)
    worst = 0.0
    for b in range(B):
        assert list(it_all[b]) == list(mi)
        assert np.array_equal(p_all[b], p_all[b % 2])
        worst = max(worst, float(np.linalg.norm(p_all[b] - refs[b % 2])))
    print("C4 B=16 (%s): worst pose error vs oracle %.2e" % (arith, worst))
    assert worst <= 1e-5
    perm = np.random.default_rng(4).permutation(B)
    assert np.array_equal(ctx.align(slots[perm], slots[perm])[0], p_all[perm])
    parts = [slots[:6], slots[6:11], slots[11:]]
    alone = [ctx.align(q, q)[0] for q in parts]
    for q in parts:
        ctx.align_enqueue(q, q)
    for q, ref in zip(parts, alone):
        assert np.array_equal(ctx.align_fetch(len(q))[0], ref)
    ctx.close()


@pytest.mark.parametrize("arith", ["fast", "exact"])
def test_dense_path_skips_the_pixels_without_depth(oracle, ellc, arith):
    """The list-free schedule (gn_fca_dense / gn_fca_dense4 in the tolerance mode, r06: gn_fca_dense_x in the exact mode; every keyframe of the batch uploaded at least nine tenths full) on a map
    with holes — a band and scattered pixels without depth, 94 % valid: the hint is not a promise, a pixel without depth contributes
    nothing; pose within 1e-5 of the oracle's (which masks them, Frame.cpp:295-301). The same keyframe thinned below nine tenths takes
    the list path: both must agree with the oracle, and with each other to the tolerance of the summation order."""
    W, H, L = 640, 480, 4
    pair = dict(synth.make_pair(W, H, seed=78, dense=True))
    rng = np.random.default_rng(5)
    d0 = pair["depth0"].copy()
    d0[100:110, :] = 0.0
    d0[rng.random(d0.shape) < 0.04] = 0.0
    pair["depth0"] = d0
    assert 0.9 < (d0 > 0).mean() < 0.97
    _, kf, cur, dm = oracle_problem(oracle, W, H, L, pair)
    p_ref, it_ref, _ = oracle.align(kf, cur, dm.depth_pyr())
    mode = ellc.ARITH_FAST if arith == "fast" else ellc.ARITH_EXACT
    ctx = gpu_problem(ellc, W, H, L, [pair], arith=mode)
    p, it, _ = ctx.align([0], [0])
    assert list(it[0]) == list(it_ref)
    assert np.linalg.norm(p[0] - p_ref) <= 1e-5
    # the list path on the same planes (a level written by itself drops the hint)
    dl, vl = ctx.keyframe_depth_level(0, 0)
    ctx.keyframe_set_depth_level(0, 0, dl, vl)
    p2, it2, _ = ctx.align([0], [0])
    assert list(it2[0]) == list(it_ref) and np.linalg.norm(p2[0] - p_ref) <= 1e-5
    assert np.linalg.norm(p2[0] - p[0]) <= 2e-6
    ctx.close()
    # ellc_ctx_set_dense_maps(1) pins the list path for a dense map: bit for bit what the list path gave above
    ctx = gpu_problem(ellc, W, H, L, [pair], arith=mode)
    ctx.set_dense_maps(1)
    p3, it3, _ = ctx.align([0], [0])
    assert np.array_equal(p3, p2) and np.array_equal(it3, it2)
    ctx.set_dense_maps(0)
    p4, _, _ = ctx.align([0], [0])
    assert np.array_equal(p4, p)
    ctx.close()


@pytest.fixture(scope="module")
def lc_batch():
    """configs[2] in the reference's shape (GlobalOptimize.cpp:566): 32 keyframes of one scene, ONE current frame."""
    return synth.make_shared_frame_batch(640, 480, 32, seed=4242)


@pytest.mark.parametrize("arith", ["exact", "fast"])
@pytest.mark.parametrize("mode", ["fca", "ica"])
def test_c2_32_keyframes_against_one_frame(oracle, ellc, lc_batch, arith, mode):
    """The loop-closure batch as the reference runs it: 32 different keyframes (own image, own depth map) aligned against one
    frame slot, 640x480, 4 levels, forward-compositional and constant-weight paths, both arithmetic modes: every one of the
    32 poses within 1e-5 of the oracle's, iteration counts equal, and the alignment recovers the rendered motion."""
    W, H, L, B = 640, 480, 4, 32
    fx, fy, cx, cy = lc_batch[0]["intrinsics"]
    cfg = ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=B, max_frames=1, max_batch=B,
                              concurrent_batches=3, arith=ellc.ARITH_FAST if arith == "fast" else ellc.ARITH_EXACT)
    ctx = ellc.Context(cfg)
    ctx.frame_upload(0, lc_batch[0]["cur_image"])
    rng = np.random.default_rng(8)
    refs = []
    for b, p in enumerate(lc_batch):
        ctx.keyframe_upload(b, p["kf_image"])
        ctx.keyframe_set_depth(b, p["depth0"], p["var0"])
        _, kf, cur, dm = oracle_problem(oracle, W, H, L, p, early_exit=0)
        if mode == "ica":
            for l in range(L):
                w = rng.uniform(0.02, 0.0625, size=(H >> l, W >> l)).astype(np.float32)
                kf.set_weights(l, w, 1)
                ctx.keyframe_set_weights(b, l, w, 1)
        refs.append(oracle.align(kf, cur, dm.depth_pyr(), loop_closure=(mode == "ica")))
    pose, iters, _ = ctx.align(np.arange(B), np.zeros(B, np.int32), mode=1 if mode == "ica" else 0)
    worst = 0.0
    for b in range(B):
        assert list(iters[b]) == list(refs[b][1])
        worst = max(worst, float(np.linalg.norm(pose[b] - refs[b][0])))
        assert np.linalg.norm(pose[b] - lc_batch[b]["xi_true"]) < 3e-3
    print("C2 shared frame (%s, %s): worst pose error vs oracle %.2e" % (mode, arith, worst))
    assert worst <= 1e-5
    ctx.close()


@pytest.mark.parametrize("arith,coalesce", [("fast", 2), ("fast", 4), ("exact", 3)])
def test_c2_coalesced_groups_equal_the_batches_alone(oracle, ellc, lc_batch, arith, coalesce):
    """cfg.coalesce at the benchmark's size (640x480, batches of 32, where the age-balanced split of the fine-level grids is
    active): batches launched side by side in one sequence give, bit for bit, what each gives when it runs alone — full
    groups, a partial group flushed by the fetch, and the pipelined pattern bench.py uses — and stay within 1e-5 of the oracle."""
    W, H, L, B = 640, 480, 4, 32
    G = coalesce + 1
    fx, fy, cx, cy = lc_batch[0]["intrinsics"]
    cfg = ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=G * B, max_frames=G, max_batch=B,
                              concurrent_batches=4 * coalesce, coalesce=coalesce, arith=ellc.ARITH_FAST if arith == "fast" else ellc.ARITH_EXACT)
    ctx = ellc.Context(cfg)
    rng = np.random.default_rng(5)
    for g in range(G):
        ctx.frame_upload(g, lc_batch[0]["cur_image"])
        order = rng.permutation(B)                       # every slot group holds the 32 keyframes in its own order
        for b in range(B):
            p = lc_batch[int(order[b])]
            ctx.keyframe_upload(g * B + b, p["kf_image"])
            ctx.keyframe_set_depth(g * B + b, p["depth0"], p["var0"])
        if g == 0:
            first = order
    kf = [np.arange(B, dtype=np.int32) + g * B for g in range(G)]
    fr = [np.full(B, g, np.int32) for g in range(G)]
    alone = [ctx.align(kf[g], fr[g]) for g in range(G)]
    # against the oracle (slot group 0, three alignments)
    for b in (0, 13, 31):
        _, okf, ocur, odm = oracle_problem(oracle, W, H, L, lc_batch[int(first[b])], early_exit=0)
        pose_ref, iters_ref, _ = oracle.align(okf, ocur, odm.depth_pyr())
        assert list(alone[0][1][b]) == list(iters_ref) and np.linalg.norm(alone[0][0][b] - pose_ref) <= 1e-5
    # full groups and one partial group (flushed by the fetch)
    for g in range(G):
        ctx.align_enqueue(kf[g], fr[g])
    for g in range(G):
        got = ctx.align_fetch(B)
        assert all(np.array_equal(x, y) for x, y in zip(got, alone[g])), g
    # the pipelined pattern: keep the queue full, fetch the oldest, enqueue the next
    n_fly = min(4 * coalesce, G)
    for s in range(n_fly):
        ctx.align_enqueue(kf[s % G], fr[s % G])
    for s in range(5 * G):
        got = ctx.align_fetch(B)
        assert all(np.array_equal(x, y) for x, y in zip(got, alone[s % G])), s
        if s + n_fly < 5 * G:
            ctx.align_enqueue(kf[(s + n_fly) % G], fr[(s + n_fly) % G])
    ctx.close()


def test_normal_equations_properties(ellc):
    """H is symmetric positive semi-definite; scaling all weights by c scales H and b by c and leaves delta unchanged (ICA)."""
    W, H, L = 320, 240, 4
    pair = synth.make_pair(W, H, seed=44)
    ctx = gpu_problem(ellc, W, H, L, [pair])
    pose = np.zeros(6, np.float32)
    g = ctx.gn_iterate(0, 0, 1, pose)
    assert np.array_equal(g["H"], g["H"].T)
    assert np.linalg.eigvalsh(g["H"].astype(np.float64)).min() > -1e-3 * np.abs(g["H"]).max()
    shp = (H >> 1, W >> 1)
    w1 = np.full(shp, 0.02, np.float32)
    ctx.keyframe_set_weights(0, 1, w1, 1)
    a = ctx.gn_iterate(0, 0, 1, pose, mode=1, it=0)
    ctx.keyframe_set_weights(0, 1, 4.0 * w1, 1)          # power-of-two scale: exact in f32
    b = ctx.gn_iterate(0, 0, 1, pose, mode=1, it=0)
    assert np.array_equal(b["H"], 4.0 * a["H"]) and np.array_equal(b["b"], 4.0 * a["b"])
    assert np.allclose(b["delta"], a["delta"], rtol=1e-4, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("concurrent", [1, 3])
def test_ica_batch_against_oracle(oracle, ellc, concurrent):
    """Loop-closure mode (constant weights) as a batch, 640x480, on the grids of both context configurations."""
    W, H, L, B = 640, 480, 4, 3
    pairs = [synth.make_pair(W, H, seed=300 + i) for i in range(B)]
    ctx = gpu_problem(ellc, W, H, L, pairs, concurrent_batches=concurrent)
    refs = []
    rng = np.random.default_rng(1)
    for b, p in enumerate(pairs):
        _, kf, cur, dm = oracle_problem(oracle, W, H, L, p)
        for l in range(L):
            w = rng.uniform(0.01, 0.0625, size=(H >> l, W >> l)).astype(np.float32)
            kf.set_weights(l, w, 1)
            ctx.keyframe_set_weights(b, l, w, 1)
        refs.append(oracle.align(kf, cur, dm.depth_pyr(), loop_closure=True)[0])
    p, it, _ = ctx.align(np.arange(B), np.arange(B), mode=1)
    for b in range(B):
        assert np.linalg.norm(p[b] - refs[b]) <= 1e-5, (b, p[b], refs[b])
    ctx.close()


def test_reference_default_size_480x270_odd_pyramid(oracle, ellc):
    """The reference's shipped configuration (ExternVariable.h:50-59): 480x270, fx=410.6014, fy=409.0370. The pyramid has odd
    sizes: stored 270,135,68,34 rows vs iterated 270,135,67,33 (Q13) — planes bit-exact on every level, pose <= 1e-5."""
    W, H, L = 480, 270, 4
    pair = synth.make_pair(W, H, seed=2017)
    pair["intrinsics"] = (410.601403, 409.037007, 240.0, 135.0)
    _, kf, cur, dm = oracle_problem(oracle, W, H, L, pair)
    ctx = gpu_problem(ellc, W, H, L, [pair])
    pose = np.array([0.003, -0.002, 0.001, 0.006, -0.003, 0.004], np.float32)
    for level in (3, 2, 1, 0):
        st = oracle.GNStepper(kf, cur, dm.depth_pyr(), level, pose, planes=True)
        ref = st.step(0)
        pl = st.get_planes()
        got = ctx.gn_iterate(0, 0, level, pose, planes=True)
        assert got["residual"].shape == (H >> level, W >> level)
        mask = kf.depth(level) > 0
        for name in ("residual", "weight", "warpedX", "warpedY"):
            assert bits_equal(got[name][mask], pl[name][mask]), (level, name)
        for k in range(6):
            assert bits_equal(got["J"][k][mask], pl["J"][k][mask]), (level, k)
        assert np.abs(got["pose"] - ref["pose"]).max() < 1e-6
        st.close()
    p_ref, it_ref, _ = oracle.align(kf, cur, dm.depth_pyr())
    p, it, _ = ctx.align([0], [0])
    assert list(it[0]) == list(it_ref)
    assert np.linalg.norm(p[0] - p_ref) <= 1e-5
    ctx.close()


def test_depth_path_at_reference_default_size(oracle, ellc):
    """observe + regularise + propagate at 480x270 with the reference intrinsics (fx != fy exercises the FX_INV quirk Q19)."""
    W, H, L = 480, 270, 4
    pair = synth.make_pair(W, H, seed=99, rot=0.005, trans=0.03)
    fx, fy, cx, cy = 410.601403, 409.037007, 240.0, 135.0
    ocfg = oracle.make_config(W, H, L, fx, fy, cx, cy)
    kf = oracle.Frame(ocfg, pair["kf_image"], 1)
    cur = oracle.Frame(ocfg, pair["cur_image"], 2)
    cur.set_pose(origin=pair["xi_true"])
    st = synth.make_depth_state(W, H, 5, pair["kf_image"], pair["idepth_true"])
    dm = oracle.DepthMap(ocfg)
    dm.set_keyframe(kf); dm.set_current(cur); dm.set_state(st)
    ctx = ellc.Context(ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=1))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.frame_upload(0, pair["cur_image"]); ctx.keyframe_from_frame(1, 0)
    ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
    dm.regularize(False); ctx.depth_regularize(False)
    dm.observe(); ctx.depth_observe(0, pair["xi_true"])
    dm.fill_holes(); ctx.depth_fill_holes()
    dm.regularize(False); ctx.depth_regularize(False)
    ref, got = dm.get_state(), ctx.depth_get_state()
    assert np.array_equal(got["valid"], ref["valid"]) and np.array_equal(got["blacklisted"], ref["blacklisted"])
    m = ref["valid"] != 0
    for f in ("invDepth", "invDepthSmoothed", "variance", "varianceSmoothed", "validity"):
        assert bits_equal(got[f][m], ref[f][m]), f
    nk = oracle.Frame(ocfg, pair["cur_image"], 3); nk.set_pose(origin=pair["xi_true"])
    dm.propagate(nk); ctx.depth_propagate(1, pair["xi_true"])
    ref, got = dm.get_state(), ctx.depth_get_state()
    assert np.array_equal(got["valid"], ref["valid"])
    m = ref["valid"] != 0
    for f in ("invDepth", "variance", "validity"):
        assert bits_equal(got[f][m], ref[f][m]), f
    ctx.close()


def test_gpu_against_committed_golden_fixtures(ellc):
    """The committed fixtures (tests/golden/*.npz, generator committed) checked directly against the device, without the
    oracle in the loop: depth stages on 96x64 and the ingest pre-pass on a 128x96 BGR frame."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as G
    gd = np.load(os.path.join(os.path.dirname(__file__), "golden", "depth_small.npz"))
    w, h, L, pair, st = G.depth_small_scene()
    fx, fy, cx, cy = pair["intrinsics"]
    ctx = ellc.Context(ellc.default_config(w, h, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=1))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.frame_upload(0, pair["cur_image"]); ctx.keyframe_from_frame(1, 0)

    def fresh():
        ctx.depth_set_keyframe(0); ctx.depth_set_state(st)

    def check(tag):
        got = ctx.depth_get_state()
        m = gd[tag + "_valid"] != 0
        assert np.array_equal(got["valid"] != 0, m), tag
        assert np.array_equal(got["blacklisted"], gd[tag + "_blacklisted"]), tag
        for f in ("invDepth", "invDepthSmoothed", "variance", "varianceSmoothed", "validity"):
            assert np.array_equal(got[f][m], gd[tag + "_" + f][m]), (tag, f)
    fresh(); ctx.depth_regularize(False); check("regularize")
    ctx.depth_observe(0, pair["xi_true"]); check("regularize_observe")
    fresh(); ctx.depth_fill_holes(); check("fill_holes")
    fresh(); ctx.depth_regularize(False); ctx.depth_propagate(1, pair["xi_true"]); check("regularize_propagate")
    ctx.close()
    gi = np.load(os.path.join(os.path.dirname(__file__), "golden", "ingest_small.npz"))
    bgr = G.ingest_small_frame()
    ctx = ellc.Context(ellc.default_config(bgr.shape[1] // 4, bgr.shape[0] // 4, 3, max_frames=1))
    kn = ctx.ingest_configure(bgr.shape[1], bgr.shape[0], G.INGEST_K[0], G.INGEST_K[1], G.INGEST_K[2], G.INGEST_K[3], G.INGEST_DIST, True)
    assert np.array_equal(kn.view(np.uint32), gi["new_camera"].view(np.uint32))
    ctx.frame_ingest_bgr(0, bgr)
    lvl0, (rows, cols) = ctx.image_level(False, 0, 0)
    assert np.array_equal(lvl0[:rows, :cols], gi["image"])
    ctx.close()

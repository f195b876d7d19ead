"""ellc_track_frame (main.cpp:330 + :368 + :499-502 as one device sequence) against the same stages called one by one."""
import os
import subprocess
import numpy as np
import pytest
from egomotion_with_local_loop_closures_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, L = 320, 240, 4


def make_ctx(ellc, pair, **kw):
    fx, fy, cx, cy = pair["intrinsics"]
    ctx = ellc.Context(ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2, **kw))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"])
    st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
    ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
    ctx.frame_upload(0, pair["cur_image"])
    return ctx


@pytest.mark.parametrize("arith", ["exact", "fast"])
@pytest.mark.parametrize("case", [(21, 0.02, 0.05), (22, 0.03, 0.08)])   # 16 iterations; 29: the state-driven schedule needs its continuation
def test_track_frame_equals_the_separate_calls(ellc, arith, case):
    seed, rot, trans = case
    pair = synth.make_pair(W, H, seed=seed, rot=rot, trans=trans)
    kw = dict(arith=ellc.ARITH_FAST) if arith == "fast" else {}
    a = make_ctx(ellc, pair, **kw)
    b = make_ctx(ellc, pair, **kw)
    for rep in range(2):   # twice: the second frame starts from the state the first left
        # one by one
        pose, iters, wgt = a.align([0], [0], save_weights=True)
        seeds = a.depth_seeds()
        pwo = ellc.concatenate_relative_pose(pose[0], np.zeros(6, np.float32))
        a.depth_observe(0, pwo); a.depth_fill_holes(); a.depth_regularize(False); a.depth_update_depth_image()
        # fused
        p2, i2, w2, s2 = b.track_frame(0, save_weights=True)
        assert np.array_equal(p2, pose[0]) and np.array_equal(i2, iters[0]) and w2 == wgt[0] and s2 == seeds
        sa, sb = a.depth_get_state(), b.depth_get_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (rep, k)
        for l in range(L):
            da, va = a.keyframe_depth_level(0, l); db, vb = b.keyframe_depth_level(0, l)
            assert np.array_equal(da, db) and np.array_equal(va, vb)
            assert np.array_equal(a.keyframe_weights(0, l)[0], b.keyframe_weights(0, l)[0])
    if case[0] == 22:
        assert int(iters.sum()) > 20 or rep   # (the continuation path was exercised on the first frame)
    a.close(); b.close()


def test_track_frame_errors(ellc):
    pair = synth.make_pair(W, H, seed=3)
    fx, fy, cx, cy = pair["intrinsics"]
    ctx = ellc.Context(ellc.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=2))
    with pytest.raises(ellc.EllcError):
        ctx.track_frame(0)           # no depth map yet
    ctx.close()


def test_driver_fused_and_unfused_write_the_same_files(tmp_path):
    """ellc_main tracks through ellc_track_frame (--fused, the default); with --no-fused every stage is its own call: identical files."""
    from test_gpu_driver import make_sequence, W as DW, H as DH, N
    frames, _ = make_sequence()
    raw = tmp_path / "frames.raw"
    raw.write_bytes(b"".join(np.ascontiguousarray(f, np.uint8).tobytes() for f in frames))
    exe = os.path.join(ROOT, "egomotion_with_local_loop_closures_amd", "csrc", "ellc_main")
    outs = []
    for name, extra in (("fused", ["--fused"]), ("unfused", ["--no-fused"])):
        d = tmp_path / name; d.mkdir()
        r = subprocess.run([exe, str(raw), str(DW), str(DH), str(N), str(d), "LC"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode()
        outs.append(((d / "poses_orig.txt").read_text(), (d / "matchframes.txt").read_text()))
    assert outs[0] == outs[1] and len(outs[0][0].strip().split("\n")) == N - 1

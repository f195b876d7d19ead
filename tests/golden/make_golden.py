"""Generates the committed golden vectors. Sources of truth:
  se3_golden.npz      scipy.linalg.expm / logm (an implementation independent of the oracle and of Eigen)
  gn_small.npz        the oracle itself (faithful-f32) on a 64x48 synthetic pair — a regression pin for the
                      restatement and the fixture the GPU path is compared with on the GPU box
Run:  python tests/golden/make_golden.py   (from the repo root; needs scipy, runs on CPU)
"""
import os
import sys
import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def hat(xi):
    w, v = xi[:3], xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


def vee(M):
    return np.array([M[2, 1], M[0, 2], M[1, 0], M[0, 3], M[1, 3], M[2, 3]])


def make_se3():
    rng = np.random.default_rng(1234)
    xi = np.concatenate([rng.normal(size=(12, 6)) * [0.02, 0.02, 0.02, 0.05, 0.05, 0.05],
                         rng.normal(size=(6, 6)) * [0.5, 0.5, 0.5, 1, 1, 1]]).astype(np.float32).astype(np.float64)
    T = np.stack([sl.expm(hat(x)) for x in xi])
    lp = np.stack([vee(np.real(sl.logm(T[i] @ T[i + 1]))) for i in range(len(xi) - 1)])
    np.savez(os.path.join(HERE, "se3_golden.npz"), xi=xi, T=T, log_of_product=lp)


def make_gn_small():
    from oracle import oracle_py as O
    from egomotion_with_local_loop_closures_amd import synth
    from helpers import oracle_problem
    w, h, L = 64, 48, 3
    pair = synth.make_pair(w, h, seed=77, rot=0.004, trans=0.008)
    mi = (4, 7, 9)
    _, kf, cur, dm = oracle_problem(O, w, h, L, pair, early_exit=0, max_iter=mi)
    out = dict(kf_image=pair["kf_image"], cur_image=pair["cur_image"], depth0=pair["depth0"], var0=pair["var0"],
               intrinsics=np.array(pair["intrinsics"], np.float32), max_iter=np.array(mi))
    pose = np.zeros(6, np.float32)
    for level in (2, 1, 0):
        st = O.GNStepper(kf, cur, dm.depth_pyr(), level, pose, planes=True)
        for it in range(2):
            r = st.step(0)
            if it == 0:
                pl = st.get_planes()
                out["L%d_residual" % level] = pl["residual"]; out["L%d_weight" % level] = pl["weight"]; out["L%d_J" % level] = pl["J"]
                out["L%d_pose_in" % level] = pose.copy()
            out["L%d_it%d_H" % (level, it)] = r["H"]; out["L%d_it%d_b" % (level, it)] = r["b"]
            out["L%d_it%d_delta" % (level, it)] = r["delta"]; out["L%d_it%d_pose" % (level, it)] = r["pose"]
        pose = r["pose"].copy()
        st.close()
    p, iters, wgt = O.align(kf, cur, dm.depth_pyr())
    out["final_pose"] = p; out["iters"] = iters
    np.savez_compressed(os.path.join(HERE, "gn_small.npz"), **out)


if __name__ == "__main__":
    make_se3()
    make_gn_small()
    print("golden vectors written to", HERE)

"""SMPL-X model data: loader for a real ``SMPLX_*.npz`` and a synthetic SMPL-X-shaped stand-in.

The licence-gated model files are absent from this build, so tests and
bench use a seeded synthetic model with the public SMPL-X dimensions and kinematic
tree (V=10475, F=20908, J=55, 486 pose-corrective rows, 51 static face landmarks).
``load_model_npz`` accepts the real file when a user supplies one; both return the
same plain dict of numpy arrays, which is what the C-ABI packer and the oracle take.

Reference call sites this serves: copenet/src/copenet/copenet_twoview.py:36-45 (ctor),
:237-241 (forward), :69,77 (.v_template, .faces).
"""
import os

import numpy as np

NUM_VERTS = 10475
NUM_FACES = 20908
NUM_JOINTS = 55
NUM_BODY_JOINTS = 21
NUM_BETAS = 10
NUM_EXPR = 10

# SMPL-X kinematic tree (public model spec)
PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)

# Vertex ids of the 21 extra joints appended after the 55 chain joints
# (upstream smplx vertex_ids['smplx'] in VertexJointSelector order: face 5, feet 6, finger tips 10).
EXTRA_JOINT_VERTS = np.array(
    [9120, 9929, 9448, 616, 6,                     # nose, reye, leye, rear, lear
     5770, 5780, 8846, 8463, 8474, 8635,           # LBigToe LSmallToe LHeel RBigToe RSmallToe RHeel
     5361, 4933, 5058, 5169, 5286,                 # l thumb index middle ring pinky
     8079, 7669, 7794, 7905, 8022], dtype=np.int64)  # r thumb index middle ring pinky
NUM_LANDMARKS = 51
NUM_OUT_JOINTS = NUM_JOINTS + len(EXTRA_JOINT_VERTS) + NUM_LANDMARKS   # 127


def make_synthetic_model(seed=4321, num_verts=NUM_VERTS, num_faces=NUM_FACES, max_bones=4, coherent=False):
    """Seeded SMPL-X-shaped model.  Geometry is body-like only in scale; all that matters for
    parity is the dimensionality, the sparsity pattern (<= max_bones weights per vertex, sparse
    joint regressor) and the real parent table."""
    rs = np.random.RandomState(seed)
    J, V = NUM_JOINTS, num_verts
    # rest joints: walk the tree with random bone offsets (~10-25 cm)
    jt = np.zeros((J, 3))
    for j in range(1, J):
        d = rs.standard_normal(3)
        d /= np.linalg.norm(d)
        jt[j] = jt[PARENTS[j]] + d * rs.uniform(0.03 if j >= 25 else 0.10, 0.06 if j >= 25 else 0.25)
    # each vertex hangs off a "main" joint
    main = rs.randint(0, J, size=V)
    if coherent:
        # vertex ids in the order of their main joint, as in the real SMPL-X mesh (neighbouring ids sit on the same limb and share
        # their bones): the default draws every vertex's joint independently, which is the WORST case for a skinning kernel that
        # gathers bone rows from LDS (16 consecutive vertices = ~16 different bones; tools/lbs_bench.py --coherent 1 measures both)
        main = np.sort(main)
    v_template = (jt[main] + rs.standard_normal((V, 3)) * 0.04).astype(np.float32)
    # skinning weights: main joint + up to 3 relatives (parent / grand-parent / a child)
    children = [[c for c in range(J) if PARENTS[c] == j] for j in range(J)]
    weights = np.zeros((V, J), np.float32)
    for v in range(V):
        j = main[v]
        cand = [j]
        if PARENTS[j] >= 0:
            cand.append(PARENTS[j])
            if PARENTS[PARENTS[j]] >= 0:
                cand.append(PARENTS[PARENTS[j]])
        if children[j]:
            cand.append(children[j][rs.randint(len(children[j]))])
        if max_bones > 4:
            # more relatives (ancestors further up, every child, siblings) so that rows with 5 .. max_bones non-zero
            # weights exist: these select the K = 8 / dynamic-K skinning kernels.  max_bones <= 4 draws exactly the
            # random numbers it always did (the default model of the tests and the bench is unchanged).
            a = cand[2] if len(cand) > 2 and PARENTS[j] >= 0 and PARENTS[PARENTS[j]] >= 0 else -1
            while a >= 0 and PARENTS[a] >= 0:
                a = PARENTS[a]
                cand.append(a)
            cand += [c for c in children[j] if c not in cand]
            if PARENTS[j] >= 0:
                cand += [c for c in children[PARENTS[j]] if c not in cand]
            cand += [c for c in range(J) if c not in cand][:max(0, max_bones - len(cand))]
        k = rs.randint(1, min(max_bones, len(cand)) + 1)
        cand = cand[:k]
        w = rs.uniform(0.1, 1.0, size=k)
        weights[v, cand] = (w / w.sum()).astype(np.float32)
    # sparse joint regressor: each joint from 24 vertices of its own cluster (or random ones)
    J_regressor = np.zeros((J, V), np.float32)
    for j in range(J):
        own = np.nonzero(main == j)[0]
        pick = own[:24] if len(own) >= 24 else rs.choice(V, 24, replace=False)
        w = rs.uniform(0.2, 1.0, size=len(pick))
        J_regressor[j, pick] = (w / w.sum()).astype(np.float32)
    shapedirs = (rs.standard_normal((V, 3, NUM_BETAS + NUM_EXPR)) * 0.02).astype(np.float32)
    posedirs = (rs.standard_normal(((J - 1) * 9, V * 3)) * 0.004).astype(np.float32)
    faces = rs.randint(0, V, size=(num_faces, 3)).astype(np.int64)
    lmk_faces_idx = rs.randint(0, num_faces, size=NUM_LANDMARKS).astype(np.int64)
    bary = rs.uniform(0.05, 1.0, size=(NUM_LANDMARKS, 3))
    lmk_bary_coords = (bary / bary.sum(1, keepdims=True)).astype(np.float32)
    extra = EXTRA_JOINT_VERTS if V > EXTRA_JOINT_VERTS.max() else rs.randint(0, V, size=21).astype(np.int64)
    # hand pose space of the model file (hands_mean{l,r} (45,), hands_components{l,r} (45,45): rows = PCA directions), from
    # its own stream so that every array above keeps the values it always had
    rh = np.random.RandomState(seed + 7)
    hands = {k: (rh.standard_normal(45) * 0.2).astype(np.float32) for k in ("hands_meanl", "hands_meanr")}
    hands.update({k: (rh.standard_normal((45, 45)) * 0.3).astype(np.float32) for k in ("hands_componentsl", "hands_componentsr")})
    return dict(hands, v_template=v_template, shapedirs=shapedirs, posedirs=posedirs,
                J_regressor=J_regressor, parents=PARENTS.copy(), lbs_weights=weights,
                faces=faces, lmk_faces_idx=lmk_faces_idx, lmk_bary_coords=lmk_bary_coords,
                extra_joint_verts=extra.copy(), synthetic=True)


def load_model_npz(path, num_betas=NUM_BETAS, num_expr=NUM_EXPR):
    """Load a real SMPL-X ``.npz`` (keys as distributed: v_template, f, shapedirs, posedirs,
    J_regressor, kintree_table, weights, lmk_faces_idx, lmk_bary_coords)."""
    d = np.load(path, allow_pickle=True)
    sd = np.asarray(d["shapedirs"], np.float32)                # (V,3,400): 300 shape + 100 expr; or (V,3,20): 10 + 10
    if sd.ndim < 3:
        sd = sd[:, :, None]
    if sd.shape[2] < 400:
        # the 10-shape / 10-expression model file: upstream smplx 0.1.28 (body_models.SMPLX.__init__) takes the
        # expression directions from columns 10:20 in that case
        num_betas, num_expr = min(num_betas, 10), min(num_expr, 10)
        expr = sd[:, :, 10:10 + num_expr]
    else:
        expr = sd[:, :, 300:300 + num_expr]
    shape = sd[:, :, :num_betas]
    if shape.shape[2] != NUM_BETAS or expr.shape[2] != NUM_EXPR:
        raise ValueError("SMPL-X file %r yields %d shape / %d expression directions; this build packs %d / %d"
                         % (path, shape.shape[2], expr.shape[2], NUM_BETAS, NUM_EXPR))
    pd = np.asarray(d["posedirs"], np.float32)                 # (V,3,486)
    posedirs = np.reshape(pd, [-1, pd.shape[-1]]).T.copy()     # (486, V*3)
    parents = np.asarray(d["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    hands = {k: np.asarray(d[k], np.float32) for k in ("hands_meanl", "hands_meanr", "hands_componentsl", "hands_componentsr")
             if k in d.files}
    return dict(hands, v_template=np.asarray(d["v_template"], np.float32),
                shapedirs=np.concatenate([shape, expr], -1),
                posedirs=posedirs,
                J_regressor=np.asarray(d["J_regressor"], np.float32),
                parents=parents,
                lbs_weights=np.asarray(d["weights"], np.float32),
                faces=np.asarray(d["f"]).astype(np.int64),
                lmk_faces_idx=np.asarray(d["lmk_faces_idx"]).astype(np.int64),
                lmk_bary_coords=np.asarray(d["lmk_bary_coords"], np.float32),
                extra_joint_verts=EXTRA_JOINT_VERTS.copy(), synthetic=False)


def find_model(model_dir, gender="neutral"):
    """Mirror of the reference ctor's path handling: a directory holding SMPLX_{GENDER}.npz or a file."""
    if model_dir is None:
        return None
    if os.path.isdir(model_dir):
        p = os.path.join(model_dir, "SMPLX_%s.npz" % gender.upper())
        return p if os.path.isfile(p) else None
    return model_dir if os.path.isfile(model_dir) else None


def sparse_skin_weights(weights):
    """(V,J) dense -> top-K (idx int32 (V,K), w float32 (V,K)); K = max non-zeros per row (>=4 padded)."""
    nnz = (weights != 0).sum(1)
    K = max(int(nnz.max()), 4)
    order = np.argsort(-np.abs(weights), axis=1, kind="stable")[:, :K]
    w = np.take_along_axis(weights, order, axis=1).astype(np.float32)
    idx = np.where(w != 0, order, 0).astype(np.int32)
    # keep bone order ascending so the summation order matches a dense left-to-right sum
    key = np.where(w != 0, idx, np.iinfo(np.int32).max)
    srt = np.argsort(key, axis=1, kind="stable")
    return np.take_along_axis(idx, srt, 1), np.take_along_axis(w, srt, 1)

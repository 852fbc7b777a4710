"""The one function the reference takes from the body-model package's `lbs` module (`from smplx import lbs`,
copenet/src/copenet/dsets/aerialpeople.py:177: `lbs.batch_rodrigues(smplpose.reshape(-1, 3))`), as one HIP launch through the
C ABI.  smplx 0.1.28 `lbs.batch_rodrigues` (the package is absent here: restated from the published source, held by
known-answer tests and the CPU oracle): angle = |r + 1e-8|, K = skew(r / angle), R = I + sin K + (1 - cos) K K."""
from .geometry import _rodrigues


def batch_rodrigues(rot_vecs, epsilon=1e-8, dtype=None):
    """(N,3) axis-angle -> (N,3,3).  `epsilon` is the published default and is what the kernel uses; `dtype` is ignored
    (float32 on the GPU)."""
    if epsilon != 1e-8:
        raise ValueError("airpose_amd.lbs.batch_rodrigues: only the published epsilon = 1e-8 is implemented")
    return _rodrigues(rot_vecs, 0, "lbs.batch_rodrigues")

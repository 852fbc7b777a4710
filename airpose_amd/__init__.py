"""airpose_amd: MI355X-native (gfx950 HIP) implementation of AirPose's two-view inference hot path.

Public mirrors of the reference interfaces:
  copenet_model.copenet / getcopenet   <- copenet/src/copenet/models/model_copenet.py
  smplx.SMPLX                          <- the smplx submodule as called by copenet_twoview.py
  geometry.rot6d_to_rotmat / perspective_projection, utils.transform_smpl
  pipeline.TwoViewInference            <- inference branch of copenet_twoview.fwd_pass_and_loss
The compute lives in libairpose_hip.so (include/airpose_hip.h); nothing here falls back to CPU.
"""
__version__ = "0.1.0"

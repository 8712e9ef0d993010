"""simple_pose_amd - the top-down pose heat-map hot path of liangheming/simple_pose, rebuilt for MI355X (gfx950).

Python here is glue (tensor ownership, stream, weight packing); every kernel lives in
simple_pose_amd/csrc/*.hip behind the C ABI of include/simple_pose_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"

"""Drop-in counterparts of the reference's `nets` package (same factories, same state_dict layout)."""
from . import pose_resnet_dconv, pose_resnet_duc  # noqa: F401

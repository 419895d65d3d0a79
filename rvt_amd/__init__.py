"""rvt_amd — MI355X (gfx950)-native recurrent-vision-transformer backbone (hot path of uzh-rpg/RVT)."""
from .backbone import RNNDetector, RNNDetectorStage, build_recurrent_backbone  # noqa: F401
from .config import AttrDict, backbone_config, modify_backbone_config  # noqa: F401

from .build import MODEL_REGISTRY, build_model  # noqa: F401
from .video_model_builder import Motionformer  # noqa: F401  (registers itself)
from .STEVE.steve import STEVE, SlotAttentionVideo  # noqa: F401  (STEVE registers itself)

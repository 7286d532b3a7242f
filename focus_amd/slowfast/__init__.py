"""Host-side mirror of the reference's operator surface for the hot path.

`focus_amd.slowfast` exposes the same import names the reference's tools use
(`slowfast.models.build_model`, `slowfast.models.MODEL_REGISTRY`, `slowfast.config.defaults.get_cfg` ...);
`install_as_slowfast()` registers it under the top-level name `slowfast` so unmodified callers such as
tools/run_net.py resolve to it.
"""
import sys


def install_as_slowfast():
    import importlib
    pkg = importlib.import_module(__name__)
    sys.modules.setdefault("slowfast", pkg)
    for sub in ("config", "config.defaults", "models", "models.build", "models.attention", "models.common",
                "models.stem_helper", "models.video_model_builder", "models.losses", "models.optimizer",
                "models.ORViT", "models.ORViT.orvit", "models.ORViT.utils", "models.ORViT.layout",
                "models.STEVE", "models.STEVE.steve", "models.STEVE.utils", "models.STEVE.transformer", "models.STEVE.dvae",
                "datasets", "datasets.utils", "datasets.transform", "utils", "utils.box_ops", "utils.distributed", "utils.misc", "utils.lr_policy", "utils.checkpoint", "utils.metrics", "utils.meters"):
        sys.modules.setdefault("slowfast." + sub, importlib.import_module(__name__ + "." + sub))
    return pkg

"""Test-infrastructure only: file-by-file loader for the reference's hot-path modules.

Runs ONLY in the build container (needs /root/reference); used by oracle/make_golden.py to
emit fixtures into tests/golden/.  Nothing under tests/ -m gpu, bench.py or smoke() imports it.

The reference package cannot be imported as a package (iopath/fvcore/torchvision/detectron2 are
absent here), so the individual source files are executed with importlib under empty namespace
packages plus plumbing-only stand-ins (logger, registry, weight-init no-ops).  The one piece of
third-party arithmetic on the path, torchvision.ops.roi_align, is absent from the image; the
reference's ObjectsCrops is bound to OUR restatement of it (oracle.roi_align) -- so RoIAlign
parity stays "unpinned"  while everything around it is the reference's code.
"""
import importlib.util
import logging
import os
import sys
import types

REF = os.environ.get("FOCUS_REFERENCE_ROOT", "/root/reference")


def _ns(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    parent, _, child = modname.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], child, mod)
    return mod


class _Registry(dict):
    def register(self):
        def deco(cls):
            self[cls.__name__] = cls
            return cls
        return deco

    def get(self, name):
        return self[name]


def load_reference(roi_align_fn):
    """Returns a dict of the reference's hot-path modules, executed from their own source files."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: " + REF)
    for n in ("slowfast", "slowfast.models", "slowfast.utils", "slowfast.models.ORViT",
              "slowfast.models.STEVE", "fvcore", "fvcore.nn", "fvcore.common", "torchvision",
              "torchvision.ops", "detectron2", "detectron2.layers"):
        _ns(n)
    # plumbing-only stand-ins
    lg = _ns("slowfast.utils.logging")
    lg.get_logger = logging.getLogger
    sys.modules["slowfast.utils"].logging = lg
    b = _ns("slowfast.models.build")
    b.MODEL_REGISTRY = _Registry()
    wi = _ns("fvcore.nn.weight_init")
    wi.c2_msra_fill = lambda *a, **k: None
    sys.modules["fvcore.nn"].weight_init = wi
    sys.modules["detectron2.layers"].ROIAlign = object
    mu = _ns("slowfast.models.utils")
    mu.round_width = lambda *a, **k: None
    sys.modules["torchvision.ops"].roi_align = roi_align_fn
    tb = _ns("torchvision.ops.boxes")
    tb.box_area = lambda bx: (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
    sys.modules["torchvision.ops"].boxes = tb

    out = {}
    out["common"] = _load("slowfast.models.common", "slowfast/models/common.py")
    out["box_ops"] = _load("slowfast.utils.box_ops", "slowfast/utils/box_ops.py")
    out["attention"] = _load("slowfast.models.attention", "slowfast/models/attention.py")
    out["layout"] = _load("slowfast.models.ORViT.layout", "slowfast/models/ORViT/layout.py")
    out["orvit_utils"] = _load("slowfast.models.ORViT.utils", "slowfast/models/ORViT/utils.py")
    out["orvit"] = _load("slowfast.models.ORViT.orvit", "slowfast/models/ORViT/orvit.py")
    out["steve_utils"] = _load("slowfast.models.STEVE.utils", "slowfast/models/STEVE/utils.py")
    out["dvae"] = _load("slowfast.models.STEVE.dvae", "slowfast/models/STEVE/dvae.py")
    out["transformer"] = _load("slowfast.models.STEVE.transformer", "slowfast/models/STEVE/transformer.py")
    out["steve"] = _load("slowfast.models.STEVE.steve", "slowfast/models/STEVE/steve.py")
    out["stem_helper"] = _load("slowfast.models.stem_helper", "slowfast/models/stem_helper.py")
    return out


def load_motionformer(mods):
    """Additionally executes the reference's model-builder file (needs the CNN helper files too)."""
    _load("slowfast.utils.weight_init_helper", "slowfast/utils/weight_init_helper.py")
    _load("slowfast.utils.distributed", "slowfast/utils/distributed.py")
    for f in ("batchnorm_helper", "nonlocal_helper", "operators", "resnet_helper", "head_helper"):
        _load("slowfast.models." + f, "slowfast/models/%s.py" % f)
    sys.modules["slowfast.models.ORViT"].ORViT = mods["orvit"].ORViT
    mods["video_model_builder"] = _load("slowfast.models.video_model_builder",
                                        "slowfast/models/video_model_builder.py")
    return mods

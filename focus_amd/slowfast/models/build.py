"""Model factory -- the drop-in boundary (mirror of slowfast/models/build.py:9-87)."""
import torch


class Registry:
    """Minimal stand-in for fvcore.common.registry.Registry (build.py:7-9): name -> class."""

    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        def deco(cls):
            assert cls.__name__ not in self._map, "%s already registered in %s" % (cls.__name__, self._name)
            self._map[cls.__name__] = cls
            return cls
        return deco if obj is None else deco(obj)

    def get(self, name):
        if name not in self._map:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return self._map[name]

    def __contains__(self, name):
        return name in self._map


MODEL_REGISTRY = Registry("MODEL")


def build_model(cfg, gpu_id=None):
    """build.py:18-87: look the model up by cfg.MODEL.MODEL_NAME, move it to the current GPU and wrap it in
    DistributedDataParallel when cfg.NUM_GPUS > 1 (one process per GPU; backend "nccl" is RCCL on ROCm)."""
    launched = torch.distributed.is_available() and torch.distributed.is_initialized()
    if torch.cuda.is_available():
        # build.py:26-29.  Under an external one-process-per-GPU launcher (torch.distributed.run) a rank may see
        # only its own device, so the check applies to the reference's own spawn model only.
        assert launched or cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    else:
        assert cfg.NUM_GPUS == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if getattr(cfg.MODEL, "LOAD_IN_PRETRAIN", "") != "":
        raise NotImplementedError("MODEL.LOAD_IN_PRETRAIN downloads weights (build.py:47-62); no network here")
    if cfg.NUM_GPUS:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)
    if cfg.NUM_GPUS > 1:
        from focus_amd.parallel import wrap_ddp
        model = wrap_ddp(model, cur_device, cfg)
    return model

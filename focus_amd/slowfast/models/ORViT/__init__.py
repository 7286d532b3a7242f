from .orvit import ORViT  # noqa: F401

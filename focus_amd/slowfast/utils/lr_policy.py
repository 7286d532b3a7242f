"""Schedules of the slot training loop (mirror of slowfast/utils/lr_policy.py:8-40; the epoch-based policies of that file
belong to the supervised loop's scheduler, which stays torch-side)."""
from focus_amd.slowfast.models.STEVE.utils import cosine_anneal, linear_warmup  # noqa: F401

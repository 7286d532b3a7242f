"""Spatial sampling of a clip together with its object boxes (mirror of slowfast/datasets/transform.py:42-294).
Every function takes frames [T,C,H,W] on ANY device (the arithmetic is plain torch: the same call augments on the GPU
when the decoded clip already lives there) and boxes [..., 4] xyxy in pixels as a numpy array or a tensor, and draws its
random numbers with the reference's own numpy calls in the reference's order."""
import math

import numpy as np
import torch


def random_short_side_scale_jitter(images, min_size, max_size, boxes=None, inverse_uniform_sampling=False):
    """transform.py:42-96: resize so that the short side is a uniformly drawn size; boxes scale with it."""
    if inverse_uniform_sampling:
        size = int(round(1.0 / np.random.uniform(1.0 / max_size, 1.0 / min_size)))
    else:
        size = int(round(np.random.uniform(min_size, max_size)))
    height, width = images.shape[2], images.shape[3]
    if (width <= height and width == size) or (height <= width and height == size):
        return images, boxes
    new_width = new_height = size
    if width < height:
        new_height = int(math.floor((float(height) / width) * size))
        if boxes is not None:
            boxes = boxes * float(new_height) / height
    else:
        new_width = int(math.floor((float(width) / height) * size))
        if boxes is not None:
            boxes = boxes * float(new_width) / width
    return torch.nn.functional.interpolate(images, size=(new_height, new_width), mode="bilinear", align_corners=False), boxes


def crop_clip_boxes(boxes, x_offset, y_offset, size):
    """transform.py:98-119: shift into the crop's frame and clip to [0, size]."""
    if isinstance(boxes, np.ndarray):
        cropped, clipf = boxes.copy(), np.clip
    else:
        cropped, clipf = boxes.clone(), torch.clip
    cropped[..., [0, 2]] = clipf(boxes[..., [0, 2]] - x_offset, 0, size)
    cropped[..., [1, 3]] = clipf(boxes[..., [1, 3]] - y_offset, 0, size)
    return cropped


def crop_boxes(boxes, x_offset, y_offset):
    """transform.py:122-138 (no clipping)."""
    cropped = boxes.copy() if isinstance(boxes, np.ndarray) else boxes.clone()
    cropped[..., [0, 2]] = boxes[..., [0, 2]] - x_offset
    cropped[..., [1, 3]] = boxes[..., [1, 3]] - y_offset
    return cropped


def random_crop(images, size, boxes=None):
    """transform.py:141-174.  (The reference returns the bare images when no crop is needed; callers unpack two values,
    so the pair is returned here in that case too.)"""
    if images.shape[2] == size and images.shape[3] == size:
        return images, boxes
    height, width = images.shape[2], images.shape[3]
    y_offset = int(np.random.randint(0, height - size)) if height > size else 0
    x_offset = int(np.random.randint(0, width - size)) if width > size else 0
    cropped = images[:, :, y_offset:y_offset + size, x_offset:x_offset + size]
    return cropped, (crop_clip_boxes(boxes, x_offset, y_offset, size) if boxes is not None else None)


def horizontal_flip(prob, images, boxes=None):
    """transform.py:177-209: x0' = W - x1 - 1, x1' = W - x0 - 1."""
    flipped = None if boxes is None else (boxes.copy() if isinstance(boxes, np.ndarray) else boxes.clone())
    if np.random.uniform() < prob:
        images = images.flip((-1))
        if images.dim() not in (3, 4):
            raise NotImplementedError("Dimension does not supported")
        width = images.shape[-1]
        if boxes is not None:
            flipped[..., [0, 2]] = width - boxes[..., [2, 0]] - 1
    return images, flipped


def uniform_crop(images, size, spatial_idx, boxes=None, scale_size=None):
    """transform.py:212-272: left / centre / right (or top / centre / bottom) crop of the test views."""
    assert spatial_idx in [0, 1, 2]
    ndim = images.dim()
    if ndim == 3:
        images = images.unsqueeze(0)
    height, width = images.shape[2], images.shape[3]
    if scale_size is not None:
        if width <= height:
            width, height = scale_size, int(height / width * scale_size)
        else:
            width, height = int(width / height * scale_size), scale_size
        images = torch.nn.functional.interpolate(images, size=(height, width), mode="bilinear", align_corners=False)
    y_offset = int(math.ceil((height - size) / 2))
    x_offset = int(math.ceil((width - size) / 2))
    if height > width:
        if spatial_idx == 0:
            y_offset = 0
        elif spatial_idx == 2:
            y_offset = height - size
    else:
        if spatial_idx == 0:
            x_offset = 0
        elif spatial_idx == 2:
            x_offset = width - size
    cropped = images[:, :, y_offset:y_offset + size, x_offset:x_offset + size]
    cropped_boxes = crop_clip_boxes(boxes, x_offset, y_offset, size) if boxes is not None else None
    if ndim == 3:
        cropped = cropped.squeeze(0)
    return cropped, cropped_boxes


def clip_boxes_to_image(boxes, height, width):
    """transform.py:275-294."""
    clipped = boxes.copy()
    clipped[:, [0, 2]] = np.minimum(width - 1.0, np.maximum(0.0, boxes[:, [0, 2]]))
    clipped[:, [1, 3]] = np.minimum(height - 1.0, np.maximum(0.0, boxes[:, [1, 3]]))
    return clipped

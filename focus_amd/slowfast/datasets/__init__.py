"""Data-side contract of the hot path: what a clip and its boxes go through between
decoding and `model(inputs, meta)` -- spatial sampling WITH boxes, pathway packing and the `orvit_bboxes` wire format.
Decoding, samplers, RandAugment and the dataset classes themselves stay the reference's."""

"""Evaluation-side operators of the StyleMesh pipeline: the multi-view consistency
("reprojection error") metric of the paper's Tab. 1, on the HIP library."""
from .reprojection import ReprojectionError, evaluate_sequence, reproject, sample_pairs, sample_pairs_det  # noqa: F401

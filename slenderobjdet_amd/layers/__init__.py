"""Operator layer: thin Python over the C-ABI HIP kernels (replaces ``slender_det/layers`` for the hot path)."""

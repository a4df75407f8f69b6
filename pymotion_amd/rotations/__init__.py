"""Rotation representations: ``quat``, ``dual_quat``, ``ortho6d`` and their ``_torch`` twins."""

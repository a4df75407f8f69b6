"""Skeleton operations (``skeleton`` = NumPy front door, ``skeleton_torch`` = torch front door)."""

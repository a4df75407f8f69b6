"""Process-wide switches of the Python front doors (the C ABI has none).

``numpy_float64_outputs`` (default True): the reference's NumPy functions hard-code float64 for some
results (``fk``: ops/skeleton.py:44, ``quat.to_matrix``: rotations/quat.py:306, ``to_euler``: :198,
``dual_quat.from_rotation_translation`` / ``from_translation``: rotations/dual_quat.py:32,51) and the
NumPy door reproduces that by up-casting the fp32 GPU results on the host.  At 2^20 frames that cast
(plus first-touch of the 2.2 GB result) is ~80 % of the door's 263 ms; set this to False to get the
GPU's float32 arrays as they are.  Functions whose reference twin keeps the input dtype are unaffected.
"""

numpy_float64_outputs = True

"""Constants shared by the host-side scene model (mirrors /root/reference/utils.py:6)."""

EPSILON = 1e-6

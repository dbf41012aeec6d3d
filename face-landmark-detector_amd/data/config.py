"""Mirror of keypoints_detector/data/config.py:6 (duplicate of networks/config.py)."""
IMAGE_ORDERING = "channels_last"

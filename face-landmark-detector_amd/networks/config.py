"""Mirror of keypoints_detector/networks/config.py:1-5: the path is channels_last only."""
IMAGE_ORDERING_CHANNELS_LAST = "channels_last"
IMAGE_ORDERING_CHANNELS_FIRST = "channels_first"

IMAGE_ORDERING = IMAGE_ORDERING_CHANNELS_LAST

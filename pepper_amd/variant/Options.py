"""Constants the inference path reads (same names as the reference's Options module:
/root/reference/pepper_variant/modules/python/Options.py:5-14,84-90)."""


class ImageSizeOptions(object):
    IMAGE_HEIGHT = 26
    IMAGE_CHANNELS = 1
    CANDIDATE_WINDOW_SIZE = 32
    TOTAL_LABELS = 28
    TOTAL_TYPE_LABELS = 3
    decoded_labels = ["HOM-REF", "HET-ALT", "HOM-ALT"]


class TrainOptions(object):
    GRU_LAYERS = 1
    HIDDEN_SIZE = 256

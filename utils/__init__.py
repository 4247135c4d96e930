"""Drop-in import path of the reference's helpers: ``from utils.utils import img_resize, load_segment``."""

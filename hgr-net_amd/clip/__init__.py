from .clip import available_models, load, tokenize  # noqa: F401
from .model import CLIP, build_model, infer_config  # noqa: F401

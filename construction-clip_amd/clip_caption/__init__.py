"""MI355X build of the prefix-caption model classes of /root/reference/CLIP_prefix_caption/train.py."""
from .model import ClipCaptionModel, ClipCaptionPrefix, GPT2LMHeadModel, KVCache, MLP, MappingType, TransformerMapper  # noqa: F401
from .generate import generate2, generate_beam  # noqa: F401
from .weights import (CaptionGeometry, GPT2_MODELS, init_caption_state_dict, init_transformer_mapper_state_dict,  # noqa: F401
                      synthetic_caption_batch)

"""Drop-in for `import clip` (openai/CLIP's public surface as used by zhuluntsai/Construction-CLIP),
executing on MI355X through libcclip_hip.so.  See clip.py / model.py."""
from .clip import available_models, load, tokenize, _transform  # noqa: F401
from . import simple_tokenizer  # noqa: F401  (attention.py:114 uses clip.simple_tokenizer.SimpleTokenizer)
from .model import CLIP, build_model  # noqa: F401
from .loss import contrastive_loss, ContrastiveLoss  # noqa: F401
from .preprocess_device import DevicePreprocess  # noqa: F401

"""Plugin surface of the reference's `model` package (model/__init__.py:1-2) for the G+D training path."""
from .model_handler import MyHandler  # noqa: F401
from .backbone import load_backbone  # noqa: F401
from .GANSurv import Generator, Discriminator, PrjDiscriminator  # noqa: F401
from .baseline_handler import BaselineHandler  # noqa: F401
from .BaseSurv import SurvNet  # noqa: F401

"""MI355X stand-in for the reference's ``dpt`` package (third_party/dpt): ``dpt.models``, ``dpt.transforms``."""
from . import models, transforms  # noqa: F401

"""limo_velo_amd — MI355X-native iterated-KF-update hot path of LIMO-Velo."""
from . import synth  # noqa: F401

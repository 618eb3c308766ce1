"""Drop-in name: ``import tokendagger as tiktoken`` resolves to the MI355X implementation (tokendagger_amd)."""
import tokendagger_amd as _impl

__version__ = _impl.__version__
__all__ = list(_impl.__all__)


def __getattr__(name):
    return getattr(_impl, name)

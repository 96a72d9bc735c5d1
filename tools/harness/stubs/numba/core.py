from . import types  # noqa

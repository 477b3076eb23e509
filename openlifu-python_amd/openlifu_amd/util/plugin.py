"""Plug-in polymorphism by class name, shared by the delay / apodization / focal-pattern / segmentation
families: ``to_dict`` tags the instance with ``"class": <ClassName>`` and ``lookup`` resolves that tag in the
family's package namespace -- the reference's JSON schema (e.g. bf/delay_methods/delaymethod.py:21-32)."""
from __future__ import annotations

import importlib


class ClassTagged:
    """Mixin: dataclass fields + a "class" tag."""

    def to_dict(self):
        d = dict(self.__dict__)
        d["class"] = type(self).__name__
        return d


def lookup(package: str, spec: dict):
    """(constructor, kwargs) for ``spec = {"class": name, **kwargs}`` inside ``package``."""
    kwargs = dict(spec)
    name = kwargs.pop("class")
    namespace = importlib.import_module(package)
    try:
        return getattr(namespace, name), kwargs
    except AttributeError:
        raise KeyError(name) from None

"""Name -> class registries (the reference's plugin mechanism).

Mirrors the behaviour of the reference registry (fdbm/util/registry.py:5-34):
``register(name)`` is a class decorator, registering a name twice warns and
replaces the old class (:17-23), ``get_by_name`` of an unknown name raises
``ValueError`` (:25-30), ``get_all_names`` lists the keys (:32-34).
"""
import warnings


class Registry:
    def __init__(self, managed_thing):
        self.managed_thing = str(managed_thing)
        self._items = {}

    def register(self, name):
        def _decorate(cls):
            if name in self._items:
                warnings.warn(
                    f"{self.managed_thing} with name '{name}' doubly registered, "
                    "old class will be replaced."
                )
            self._items[name] = cls
            return cls

        return _decorate

    def get_by_name(self, name):
        try:
            return self._items[name]
        except KeyError:
            raise ValueError(f"{self.managed_thing} with name '{name}' unknown.") from None

    def get_all_names(self):
        return list(self._items)


BridgeRegistry = Registry("Bridge")
BackboneRegistry = Registry("Backbone")
PredictorRegistry = Registry("Predictor")
CorrectorRegistry = Registry("Corrector")

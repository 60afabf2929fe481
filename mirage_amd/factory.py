"""Decorator registry used for `model_factory[name](...)` (same contract as mutils/factory.py:5-36)."""
from typing import Any, Dict, Optional, Tuple


def get_factory_adder() -> Tuple[Any, Dict[str, Any]]:
    registry: Dict[str, Any] = {}

    def _register(obj: Any, name: Optional[str] = None) -> Any:
        registry[name if name is not None else obj.__name__] = obj
        return obj

    def add(obj_or_name: Any, name: Optional[str] = None) -> Any:
        if callable(obj_or_name):
            return _register(obj_or_name)
        return lambda obj: _register(obj, obj_or_name)

    return add, registry

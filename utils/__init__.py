"""Import shim for ``utils.optimizer``.  Every other ``utils.*`` module
(``utils.evaluate``, ``utils.dataloader`` ...) keeps resolving to the
reference's own package further down ``sys.path``."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

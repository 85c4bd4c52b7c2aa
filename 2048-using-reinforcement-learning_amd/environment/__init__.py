"""`environment` package of the drop-in layout (game_2048.py is replaced; anything else resolves to the reference)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

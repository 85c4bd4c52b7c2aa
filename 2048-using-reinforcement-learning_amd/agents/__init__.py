"""`agents` package of the drop-in layout. Only beam_search_agent.py is replaced here; every other module of the
reference's `agents` package (ppo_agent.py, hybrid.py) keeps resolving to the reference checkout further down sys.path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

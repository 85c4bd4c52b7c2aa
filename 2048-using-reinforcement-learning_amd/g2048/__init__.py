"""g2048 -- MI355X-native batched 2048 rollout / beam-search engine (host side).

`ops` holds one tensor-level wrapper per C-ABI entry point (include/g2048.h);
`VecGame2048` and `BatchedBeamSearch` are the batched front-ends; the drop-in
look-alikes of the reference's classes live where the reference keeps them:
`environment.game_2048.Game2048Env` and `agents.beam_search_agent.BeamSearchAgent`
(put this package directory on sys.path instead of the reference checkout).
"""
from . import _lib, ops                      # noqa: F401
from .vec import VecGame2048, BatchedBeamSearch   # noqa: F401
from .evaluate import evaluate_beam_search, evaluate_beam_search_sharded, save_moveset, save_game_data   # noqa: F401
from .rollout import RolloutCollector, masked_sample  # noqa: F401

__all__ = ["ops", "VecGame2048", "BatchedBeamSearch", "evaluate_beam_search", "evaluate_beam_search_sharded", "save_moveset",
           "save_game_data", "RolloutCollector", "masked_sample"]

"""Drop-in for the reference's agents/beam_search_agent.py: same class name, constructor, get_action
contract, no-op remember/update, JSON save/load. The search itself runs on the GPU
(`g2048_beam_get_action`, one wavefront per game); for many games at once use
`g2048.BatchedBeamSearch`. The reference's quirks are reproduced by default (its DOWN move returns the
180-degree-rotated board; the phase is fixed from the root) -- see DESIGN.md."""
import json
import os
import random

import numpy as np
import torch

from g2048 import _lib as L
from g2048 import ops


class BeamSearchAgent:
    def __init__(self, beam_width=10, search_depth=15, seed=None, device="cuda"):      # reference :13-30
        if not (1 <= int(beam_width) <= L.BEAM_MAX_WIDTH):
            raise ValueError("BeamSearchAgent: beam_width must be in 1..%d on the MI355X engine" % L.BEAM_MAX_WIDTH)
        self.beam_width = beam_width
        self.search_depth = search_depth
        self.action_names = {0: "LEFT", 1: "UP", 2: "RIGHT", 3: "DOWN"}
        self.early_game_threshold = 512
        self.mid_game_threshold = 1024
        self.device = torch.device(device)
        self.seed = random.getrandbits(63) if seed is None else int(seed)
        self._calls = 0
        self._spawns = 0
        self._init_patterns()

    def _init_patterns(self):                                                           # reference :32-69
        """The reference's public pattern tables, kept as attributes for API parity. Only snake_patterns[0]
        takes part in the score (reference :369); it lives in the kernel as four v_dot4 weight words."""
        self.snake_patterns = [np.array([[15, 14, 13, 12], [8, 9, 10, 11], [7, 6, 5, 4], [0, 1, 2, 3]]),
                               np.arange(15, -1, -1).reshape(4, 4)]
        base = np.arange(4)
        self.gradients = [4 + base[:, None] - base[None, :], 7 - base[:, None] - base[None, :]]
        self.corners = [(0, 0), (0, 3), (3, 0), (3, 3)]

    def _lean_buffers(self):
        """What one decision needs, allocated once: a pinned host block the kernels read the state from and write the answer to
        (device-visible memory: no copy in either direction), and the packed root on the device.
        host block: [0,64) int32 tiles in | [64] action out | [65] caller mask in | [68,72) f32 prob out."""
        self._hostblk = torch.zeros(80, dtype=torch.uint8).pin_memory()
        self._hb = self._hostblk.numpy()
        self._hb_tiles = self._hb[0:64].view(np.int32)
        self._hb_prob = self._hb[68:72].view(np.float32)
        self._hb_prob_bits = self._hb[68:72].view(np.uint32)
        self._root = torch.zeros((1, 16), dtype=torch.uint8, device=self.device)
        base = self._hostblk.data_ptr()
        self._p_tiles, self._p_action, self._p_mask, self._p_prob = base, base + 64, base + 65, base + 68
        lib = L.lib()
        self._f_pack, self._f_beam = lib.g2048_pack_i32, lib.g2048_beam_get_action

    def get_action(self, state, valid_moves=None):                                      # reference :71-181
        """One decision = two launches and no copy: g2048_pack_i32 reads the state from pinned host memory, g2048_beam_get_action
        writes action and probability back into it, and the host polls the two (each a single store, preset to values no decision
        produces) instead of synchronising the stream."""
        if self.device.type != "cuda":
            raise RuntimeError("BeamSearchAgent: needs a ROCm device; there is no CPU path")
        if getattr(self, "_hostblk", None) is None:
            self._lean_buffers()
        hb = self._hb
        self._hb_tiles[:] = np.asarray(state, dtype=np.int32).reshape(16)
        hb[64] = 0xFF                               # no decision is action 255 ...
        self._hb_prob_bits[0] = 0x7FC00001          # ... or this NaN
        mask_ptr = None
        if valid_moves is not None:
            hb[65] = sum(int(bool(v)) << a for a, v in enumerate(list(valid_moves)[:4]))
            mask_ptr = self._p_mask
        stream = torch.cuda.current_stream(self.device)
        sp = stream.cuda_stream
        L.call(self.device, self._f_pack, self._p_tiles, self._root.data_ptr(), 1, sp)
        L.call(self.device, self._f_beam, self._root.data_ptr(), mask_ptr, self._p_action, self._p_prob, None, int(self.beam_width),
               int(self.search_depth), int(self.early_game_threshold), int(self.mid_game_threshold), L.u64(self.seed),
               L.u64(self._calls), 0, 1, 0, sp)
        self._calls += 1
        bits, spins = self._hb_prob_bits, 0
        while hb[64] == 0xFF or bits[0] == 0x7FC00001:
            spins += 1
            if spins > 2000000:                     # (~1 s: let the runtime say what is wrong)
                stream.synchronize()
                if hb[64] == 0xFF or bits[0] == 0x7FC00001:
                    raise RuntimeError("g2048: the decision never arrived")
        return int(hb[64]), float(self._hb_prob[0])

    # -- the reference's per-board helpers, for scripts that call them directly (one board, one small launch each) ----------
    def _codes(self, board):
        tiles = torch.as_tensor(np.ascontiguousarray(board, dtype=np.int32).reshape(1, 16), device=self.device)
        return ops.pack(tiles)

    def _determine_game_phase(self, max_tile):                                          # reference :271-278
        return "early" if max_tile < self.early_game_threshold else "mid" if max_tile < self.mid_game_threshold else "late"

    def _check_valid_moves(self, board):                                                # reference :183-192 (DOWN quirk included)
        m = int(ops.valid_moves(self._codes(board), agent_semantics=True).item())
        return [bool((m >> a) & 1) for a in range(4)]

    def _fast_evaluate(self, board, game_phase=None):                                   # reference :280-314 (the phase is ignored there too)
        return float(ops.evaluate(self._codes(board), L.EVAL_FAST).item())

    def _evaluate_state(self, board, game_phase):                                       # reference :316-373
        phase = torch.tensor([("early", "mid", "late").index(game_phase)], dtype=torch.uint8, device=self.device)
        return float(ops.evaluate(self._codes(board), L.EVAL_FULL, phase).item())

    def _calculate_corner_bonus(self, board):                                           # reference :375-385
        return float(ops.evaluate(self._codes(board), L.EVAL_CORNER_BONUS).item())

    def _calculate_merge_potential(self, board):                                        # reference :387-403
        return float(ops.evaluate(self._codes(board), L.EVAL_MERGE_POTENTIAL).item())

    def _one_board_op(self, board, op, action=0, index=0):
        """g2048_env_step on a scratch copy of `board`: returns (tiles int32[4,4], score word, flags)."""
        codes = self._codes(board)
        score = torch.zeros(1, dtype=torch.int32, device=self.device)
        record = torch.zeros(L.ENV_RECORD_BYTES, dtype=torch.uint8, device=self.device)
        ops.env_step(codes, score, record, self.seed, index, 0, action, op)
        h = record.cpu().numpy()
        return h[0:64].view(np.int32).reshape(4, 4).copy(), int(h[64:68].view(np.int32)[0]), int(h[68])

    def _make_move(self, board, action):                                                # reference :194-258
        """(new_board, score_gained, move_was_valid) with the reference's semantics: DOWN returns the 180-degree-rotated
        result (:209-210 vs :251-253), and an action outside 1..3 slides LEFT (no pre / post transform applies to it)."""
        a = int(action)
        tiles, gained, flags = self._one_board_op(board, L.ENV_OP_MOVE_AGENT, a if a in (1, 2, 3) else 0)
        return tiles.astype(np.asarray(board).dtype, copy=False), gained, bool(flags & L.FLAG_VALID)

    def _add_random_tile(self, board):                                                  # reference :260-269: IN PLACE, returns None
        tiles, _, _ = self._one_board_op(board, L.ENV_OP_SPAWN, index=self._spawns)
        self._spawns += 1
        board[...] = tiles.reshape(np.shape(board))

    def remember(self, *args):                                                          # reference :405-407
        pass

    def update(self):                                                                   # reference :409-411
        pass

    def save(self, path):                                                               # reference :413-449
        config = {
            "beam_width": self.beam_width,
            "search_depth": self.search_depth,
            "early_game_threshold": self.early_game_threshold,
            "mid_game_threshold": self.mid_game_threshold,
        }
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(config, f, indent=4)
        print(f"Beam Search configuration saved to {path}")
        readme_path = os.path.join(os.path.dirname(path),
                                   f"beam_search_config_readme_{self.beam_width}_{self.search_depth}.txt")
        with open(readme_path, "w") as f:
            f.write("Beam Search Agent Configuration\n==============================\n\n")
            f.write(f"Beam Width: {self.beam_width}\nSearch Depth: {self.search_depth}\n")
            f.write(f"Early Game Threshold: {self.early_game_threshold}\nMid Game Threshold: {self.mid_game_threshold}\n")
            f.write(f"\nSaved at: {path}\n")
            f.write("\nThis configuration achieved good results in training.\n")
            f.write("To recreate this agent, use:\n")
            f.write(f"agent = BeamSearchAgent(beam_width={self.beam_width}, search_depth={self.search_depth})")

    @classmethod
    def load(cls, path):                                                                # reference :451-478
        with open(path, "r") as f:
            config = json.load(f)
        agent = cls(beam_width=config.get("beam_width", 10), search_depth=config.get("search_depth", 15))
        if "early_game_threshold" in config:
            agent.early_game_threshold = config["early_game_threshold"]
        if "mid_game_threshold" in config:
            agent.mid_game_threshold = config["mid_game_threshold"]
        print(f"Beam Search configuration loaded from {path}")
        return agent

"""Drop-in for the reference's environment/game_2048.py: same class name, methods, return types.

One board living on the GPU and stepped by the same device arithmetic that steps millions. It exists so
that scripts written against the reference (`train.py`-shaped loops, `agents/ppo_agent.py`) run unchanged;
for throughput use `g2048.VecGame2048`. A `train.py` iteration (`get_valid_moves()` + `step()`, train.py:55-75)
costs ONE launch and ONE synchronisation: `g2048_env_step` takes the action by value, steps the board in place and
writes one 80-byte record -- state as int32 tiles, score, flags, the NEXT state's valid-move mask, f64 reward -- that
all host mirrors are refreshed from, so `get_valid_moves()` is a cache read. Differences from the reference, all deliberate:
  * tile spawns come from the engine's counter RNG, not from Python's global `random`
    (the seed defaults to one draw from `random`, so `random.seed(k)` still pins a run);
  * size != 4 raises (the reference's agents hard-code 4x4: beam_search_agent.py:69,378);
  * step(action) with an action outside 0..3 is, as in the reference (`_execute_move` has no branch for it, :97-114), a
    move that changes nothing: invalid, no spawn, the invalid-move reward (the batched front-ends use the low two bits of the
    action byte unless asked otherwise: G2048_STEP_NOOP_ACTIONS).
"""
import random

import numpy as np
import torch

from g2048 import _lib as L
from g2048 import ops


class Game2048Env:
    ACTIONS = {0: "LEFT", 1: "UP", 2: "RIGHT", 3: "DOWN"}      # reference :11-16

    # where the 80-byte step record lands: "host" = the kernel writes it straight into pinned (device-visible) host memory, so
    # an iteration is one launch + one stream synchronisation; "device" = device memory + one device->host copy
    RECORD = "host"

    def __init__(self, size=4, seed=None, device="cuda", record=None):
        if size != 4:
            raise ValueError("Game2048Env: only size=4 is supported by the MI355X engine")
        self.size = size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Game2048Env: needs a ROCm device; there is no CPU path")
        L.lib()
        self.seed = random.getrandbits(63) if seed is None else int(seed)
        self._boards = torch.zeros((1, 16), dtype=torch.uint8, device=self.device)
        self._scores = torch.zeros(1, dtype=torch.int32, device=self.device)
        # ONE 80-byte record receives everything an iteration produces (g2048_env_step): [0:64) tiles int32[16] | [64:68) score
        # int32 | [68] flags | [69] valid-move mask of the new state | [72:80) reward float64
        self._host = torch.zeros(L.ENV_RECORD_BYTES, dtype=torch.uint8).pin_memory()
        self._record_on_host = (record or self.RECORD) == "host"
        self._rec = self._host if self._record_on_host else torch.zeros(L.ENV_RECORD_BYTES, dtype=torch.uint8, device=self.device)
        self._h = self._host.numpy()
        self._h16 = self._h.view(np.uint16)         # [35] = bytes 70..71: the token the kernel writes last
        self._tok = 0
        self._t = 0
        self._epoch = 0
        self._spawns = 0            # add_new_tile() calls made directly (their draws are a stream of their own)
        self._mask = None           # valid-move mask of the current board, if the last record still describes it
        self.highest_tile = 0
        self.reset()

    # -- state mirrors (host copies refreshed by every device call) ----------
    @property
    def board(self):
        return self._board_np

    @board.setter
    def board(self, value):
        tiles = torch.as_tensor(np.ascontiguousarray(value, dtype=np.int32).reshape(1, 16), device=self.device)
        ops.pack(tiles, out=self._boards)
        self._board_np = np.array(value, dtype=np.int32).reshape(4, 4).copy()
        self._mask = None           # the cached mask described the old board

    @property
    def score(self):
        return self._score

    @score.setter
    def score(self, value):
        self._score = value
        self._scores.fill_(int(value))

    def _run(self, op, index, action=0):
        """One g2048_env_step launch and the wait for its record; refreshes the host mirrors. Returns (flags, reward).
        With the record in pinned host memory the wait is a poll of the record's token (written last by the kernel, behind a
        system-scope fence): no stream synchronisation, whose wake-up alone costs more than the launch."""
        if self._record_on_host:
            tok = self._tok = self._tok % 65535 + 1
            ops.env_step(self._boards, self._scores, self._rec, self.seed, index, 0, action, op | (tok << L.ENV_TOKEN_SHIFT))
            h16, spins = self._h16, 0
            while h16[35] != tok:
                spins += 1
                if spins > 2000000:         # (~1 s: something is wrong with the launch -- let the runtime say what)
                    torch.cuda.current_stream(self.device).synchronize()
                    if h16[35] != tok:
                        raise RuntimeError("g2048: the env record never arrived")
        else:
            ops.env_step(self._boards, self._scores, self._rec, self.seed, index, 0, action, op)
            stream = torch.cuda.current_stream(self.device)
            self._host.copy_(self._rec, non_blocking=True)
            stream.synchronize()
        h = self._h
        self._board_np = h[0:64].view(np.int32).reshape(4, 4).copy()
        self._score = np.int32(h[64:68].view(np.int32)[0])
        self._mask = int(h[69])
        return int(h[68]), np.float64(h[72:80].view(np.float64)[0])

    # -- reference API ---------------------------------------------------------
    def reset(self):                                           # reference :29-48
        self._run(L.ENV_OP_RESET, self._epoch)
        self._epoch += 1
        self._score = 0
        self.game_over = False
        self.highest_tile = np.max(self._board_np)
        return self.get_state()

    def get_state(self):                                       # reference :50-57
        return self._board_np.flatten()

    def get_valid_moves(self):                                 # reference :69-95
        if self._mask is None:      # `board` was assigned since the last step: ask the device once (no move, no draw)
            self._run(L.ENV_OP_PEEK, 0)
        m = self._mask
        return [bool((m >> a) & 1) for a in range(4)]

    def step(self, action):                                    # reference :170-210
        a = int(action)
        # anything but 0..3 moves nothing, as the reference's _execute_move (:97-114)
        flags, reward = self._run(L.ENV_OP_STEP, self._t, a if a in (0, 1, 2, 3) else 255)
        self._t += 1
        self.game_over = bool(flags & L.FLAG_DONE)
        current_highest = np.max(self._board_np)
        if current_highest > self.highest_tile:
            self.highest_tile = current_highest
        return self.get_state(), reward, self.game_over, {
            "score": self._score,
            "valid_move": bool(flags & L.FLAG_VALID),
            "highest_tile": self.highest_tile,
        }

    # -- the pieces of a step, for scripts that drive them directly (one small launch each, as step()) -----------------
    def add_new_tile(self):                                    # reference :59-67
        """A 2 (90 %) or a 4 on a random empty cell; nothing on a full board. The draw is (seed, STEP, k, counter 1) for the
        k-th direct call -- never a draw step() itself uses."""
        self._run(L.ENV_OP_SPAWN, self._spawns)
        self._spawns += 1

    def _execute_move(self, action):                           # reference :97-114: the move alone, no new tile
        a = int(action)
        self._run(L.ENV_OP_MOVE, 0, a if a in (0, 1, 2, 3) else 255)

    def _move_left(self):                                      # reference :116-168
        flags, _ = self._run(L.ENV_OP_MOVE, 0, 0)
        return bool(flags & L.FLAG_VALID)

    def simulate_move(self, state, action):                    # reference :341-387
        """All (next_state, reward, done) successors of `action` on `state`, as the reference lists them
        (it does not touch the env's own board or score)."""
        tiles = torch.as_tensor(np.ascontiguousarray(state, dtype=np.int32).reshape(1, 16), device=self.device)
        hi = int(self.highest_tile)
        hc = torch.tensor([hi.bit_length() - 1 if hi > 0 else 0], dtype=torch.uint8, device=self.device)
        act = torch.tensor([int(action) & 3], dtype=torch.uint8, device=self.device)
        succ, reward, done, count = ops.simulate_move(ops.pack(tiles), act, hc)
        k = int(count.item())
        states = ops.unpack(succ[0, :k].contiguous()).cpu().numpy()
        rewards = reward[0, :k].cpu().numpy()
        dones = done[0, :k].cpu().numpy()
        return [(states[i], np.float64(rewards[i]), bool(dones[i])) for i in range(k)]

    def is_game_over(self):                                    # reference :279-288
        return not any(self.get_valid_moves())

    def _evaluate_pattern(self):                               # reference :313-339 (no caller there; kept for the API)
        return float(ops.evaluate(self._boards, L.EVAL_PATTERN).item())

    def render(self, mode="human"):                            # reference :290-311 (host-side text)
        if mode == "human":
            print("-" * (5 * self.size + 1))
            for row in self._board_np:
                print("|" + "".join("    |" if int(t) == 0 else "%4d|" % int(t) for t in row))
                print("-" * (5 * self.size + 1))
            print("Score: %s" % self._score)
            print("Highest Tile: %s" % self.highest_tile)
            print()

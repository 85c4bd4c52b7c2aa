"""Batched front-ends: thousands to millions of independent games resident on one GPU."""
import torch

from . import _lib as L
from . import ops


class _StepInfo(dict):
    """info dict of VecGame2048.step, computed on access (a step itself launches one kernel and nothing else).
    Keys as the reference's (environment/game_2048.py:206-210): score, valid_move, highest_tile. Views of the
    env's live buffers: read them before the next step."""

    def __init__(self, env):
        super().__init__()
        self._env = env

    def __missing__(self, key):
        e = self._env
        if key == "score":
            v = e.scores
        elif key == "valid_move":
            v = (e.flags & L.FLAG_VALID).bool()
        elif key == "highest_tile":
            code = (e.flags >> L.FLAG_MAXCODE_SHIFT).int()
            v = torch.where(code > 0, torch.ones_like(code) << code, torch.zeros_like(code))
        else:
            raise KeyError(key)
        self[key] = v
        return v

    def keys(self):
        return ["score", "valid_move", "highest_tile"]


class VecGame2048:
    """n independent Game2048Env instances living in HBM (reference environment/game_2048.py:4-210).

    State: boards uint8 (n,16) log2 codes, scores int32 (n,). All draws are keyed by
    (seed, global board id = id_base + i, step counter), so a run is reproducible and independent of
    how boards are sharded over GPUs (one VecGame2048 per rank with its own id_base).

    chains=C (default 1) steps the boards as C independent sub-batches -- contiguous slices, each a g2048_step launch of
    its own on its own stream, ordered only behind the same slice's previous step. Boards are independent, so the result
    is the single launch's bit for bit (draws are keyed by global board id); what changes is the launch form: one
    chain's launch head and drain overlap the other chains' arithmetic, which a single launch per step cannot do
    (12.4 -> 10.2-10.7 us per 1 Mi-board step with two chains on two streams). `step(..., join=False)`
    leaves the chains open across steps -- that is where the overlap comes from; see `step` and `join`.
    """

    ACTIONS = {0: "LEFT", 1: "UP", 2: "RIGHT", 3: "DOWN"}      # game_2048.py:11-16

    def __init__(self, n, device="cuda", seed=0x2048, id_base=0, auto_reset=False, reward_f64=False, chains=1):
        self.n = int(n)
        self.chains = int(chains)
        if self.chains < 1:
            raise ValueError("g2048: chains must be at least 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("g2048: VecGame2048 needs a ROCm device (got %s); there is no CPU path" % self.device)
        L.lib()
        self.seed, self.id_base = int(seed), int(id_base)
        self.auto_reset, self.reward_f64 = bool(auto_reset), bool(reward_f64)
        self.boards = torch.zeros((self.n, 16), dtype=torch.uint8, device=self.device)
        self._spare = torch.empty_like(self.boards)
        self.scores = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.reward = torch.empty(self.n, dtype=torch.float64 if reward_f64 else torch.float32, device=self.device)
        self.flags = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        self.t = 0            # step counter (RNG index)
        self.epoch = 0        # reset counter (RNG index)
        self._chains = ops.StepChains(self.n, self.chains, self.device)
        self.chain_bounds = self._chains.bounds
        if len(self.chain_bounds) > 1:
            self._chains.keep_alive(self.boards, self._spare, self.scores, self.reward, self.flags)
        # Launch arguments of step(), prepared once: the state tensors never move (boards / _spare swap roles, nothing is
        # reallocated), so a step is one foreign call per chain with the action pointer, the step index and the stream filled in --
        # no tensor checks, no slicing, no stream context in the loop (tools/vec_rate.py).
        self._fn = L.lib().g2048_step
        self._prepare()
        self._done_u8 = torch.zeros(self.n, dtype=torch.uint8, device=self.device)      # step()'s `done`: one elementwise launch into a
        self._done = self._done_u8.view(torch.bool)                                      # buffer of the env (a live view, like `info`)
        self.reset()

    def _prepare(self):
        """(Re)build the prepared launch arguments from the state tensors as they are now: once in the constructor, and again
        should a caller have replaced one of them (`env.boards = ...`), which step() notices by the pointers."""
        for name, t, dt in (("boards", self.boards, torch.uint8), ("scores", self.scores, torch.int32), ("flags", self.flags, torch.uint8)):
            L.require_device_tensor(t, dt, (16,) if name == "boards" else None, name)
            if t.shape[0] != self.n:
                raise ValueError("g2048: %s must hold %d boards" % (name, self.n))
        L.require_device_tensor(self.reward, torch.float64 if self.reward_f64 else torch.float32, None, "reward")
        if self._spare.shape != self.boards.shape or self._spare.device != self.boards.device:
            self._spare = torch.empty_like(self.boards)
        self._opts = (L.STEP_REWARD_F64 if self.reward_f64 else 0) | (L.STEP_AUTO_RESET if self.auto_reset else 0)
        # chains of at least 256 Ki boards: two boards per lane (G2048_STEP_TUNE 2), so that the chains' launches -- 4 wavefronts per
        # SIMD each -- are resident together instead of taking turns; smaller chains keep the library's choice
        if len(self.chain_bounds) > 1 and min(hi - lo for lo, hi in self.chain_bounds) >= (1 << 18):
            self._opts |= 2 << 8
        rb = self.reward.element_size()
        a, b = self.boards.data_ptr(), self._spare.data_ptr()
        self._ptr_a, self._ptr_b = a, b
        self._ptr_state = (self.scores.data_ptr(), self.reward.data_ptr(), self.flags.data_ptr())
        self._lanes = []
        for lo, hi in self.chain_bounds:
            fixed = (self.scores.data_ptr() + 4 * lo, self.reward.data_ptr() + rb * lo, self.flags.data_ptr() + lo)
            self._lanes.append((lo, hi - lo, (a + 16 * lo, b + 16 * lo), fixed, L.u64(self.id_base + lo)))
        self._seed64 = L.u64(self.seed)
        if len(self.chain_bounds) > 1:
            self._chains.keep_alive(self.boards, self._spare, self.scores, self.reward, self.flags)

    # ---- independent sub-batch chains (ops.StepChains) -----------------------------------------------------------------
    def fence(self):
        """Open chains wait for everything queued on the current stream so far (e.g. an actions tensor produced after the
        chains were opened). Not needed for inputs that existed before the first `step(..., join=False)`."""
        self._chains.fence()

    def join(self):
        """Order the stream the chains were opened on behind every chain's last step; the state tensors are then ordinary
        tensors of that stream again. No-op when nothing is open; no host synchronisation."""
        self._chains.join()

    def reset(self):
        """All boards: empty + two spawns, score 0. Returns the boards tensor (codes)."""
        self.join()
        ops.reset(self.n, self.seed, self.epoch, self.id_base, boards=self.boards, scores=self.scores)
        self.epoch += 1
        self.flags.zero_()
        return self.boards

    def load(self, boards, scores=None):
        """Overwrite the state (e.g. synthetic benchmark boards)."""
        L.require_device_tensor(boards, torch.uint8, (16,), "boards")
        self.join()
        self.boards.copy_(boards)
        if scores is None:
            self.scores.zero_()
        else:
            self.scores.copy_(scores)
        return self.boards

    def step(self, actions=None, join=True):
        """actions uint8 (n,), or None for a random playout step (uniform actions drawn inside the kernel, the same ones
        `random_actions()` returns for this step). Returns (boards, reward, done(bool), info) -- tensors, no host sync.
        info: score, valid_move, highest_tile (as in game_2048.py:206-210). All four are views of the env's live buffers:
        read (or clone) them before the next step. Host cost: one foreign call per chain plus one elementwise launch for `done`
        (tools/vec_rate.py: ~13 us per step() with one chain, the same with two chains and join=False).

        chains > 1: one launch per chain, chain c on its own stream behind chain c's previous step. join=True (default)
        closes the chains before returning: the results are ordered on the current stream like a single launch's -- and every
        step pays a fork and a join (~35 us per step at 1 Mi boards against ~13 us with one chain: chains only pay with
        join=False, ~11 us per step).
        join=False leaves them open and returns None: further `step(..., join=False)` calls queue behind their own chain
        only (capture such a loop in a hipGraph, or keep the host ahead of the GPU, and the chains overlap); call `join()`
        before reading `boards` / `reward` / `flags` / `scores` on the current stream. Inputs of an open chain must have been
        queued before the chains were opened, or be handed over with `fence()`."""
        if actions is None:
            act, opts = 0, self._opts | L.STEP_RANDOM_ACTIONS
        else:
            if (not isinstance(actions, torch.Tensor) or actions.dtype != torch.uint8 or actions.device != self.boards.device
                    or actions.dim() != 1 or actions.shape[0] != self.n or not actions.is_contiguous()):
                L.require_device_tensor(actions, torch.uint8, None, "actions")           # (says what is wrong)
                raise ValueError("g2048: actions must be a contiguous uint8 tensor of one action per board on the env's device")
            act, opts = actions.data_ptr(), self._opts
        bp = self.boards.data_ptr()
        if (bp != self._ptr_a and bp != self._ptr_b) or self._spare.data_ptr() not in (self._ptr_a, self._ptr_b) or \
                (self.scores.data_ptr(), self.reward.data_ptr(), self.flags.data_ptr()) != self._ptr_state:
            self.join()
            self._prepare()              # a state tensor was replaced from outside
            bp = self.boards.data_ptr()
        src = 0 if bp == self._ptr_a else 1          # which of the two board buffers holds the state
        many = len(self._lanes) > 1
        if many:
            if actions is not None:
                self._chains.hold(actions)
            self._chains.fork()
        t = L.u64(self.t)
        if torch.cuda.current_device() != self._dev_index():
            with torch.cuda.device(self.device):
                rc = self._launch(act, opts, src, t, many)
        else:
            rc = self._launch(act, opts, src, t, many)
        if rc != L.OK:
            L.check(rc)
        self.boards, self._spare = self._spare, self.boards
        self.t += 1
        if not join and many:
            return None
        self.join()
        torch.bitwise_and(self.flags, L.FLAG_DONE, out=self._done_u8)
        return self.boards, self.reward, self._done, _StepInfo(self)

    def _dev_index(self):
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _launch(self, act, opts, src, t, many):
        fn, seed = self._fn, self._seed64
        for c, (lo, n, boards, fixed, id_base) in enumerate(self._lanes):
            stream = self._chains.stream(c).cuda_stream if many else torch.cuda.current_stream(self.device).cuda_stream
            rc = fn(boards[src], (act + lo) if act else None, boards[1 - src], fixed[0], fixed[1], fixed[2], seed, t, id_base, n, opts, stream)
            if rc != L.OK:
                return rc
        return L.OK

    def random_playout(self, steps, want_rewards=False, want_flags=False, want_episodes=False):
        """`steps` random-playout steps of every board in ONE launch (g2048_step_many: the boards stay in registers between
        the steps): exactly `steps` calls of `step()` without actions, minus the per-step round trips. Returns (boards, last
        flags, reward stream (steps, n) or None, flags stream (steps, n) or None, episodes finished per board or None);
        `self.reward` is not updated (ask for the stream)."""
        steps = int(steps)
        self.join()
        out, flags, rewards, fstream, episodes = ops.step_many(
            self.boards, self.scores, self.seed, self.t, steps, self.id_base, out=self._spare, flags=self.flags,
            reward_f64=self.reward_f64, auto_reset=self.auto_reset, want_rewards=want_rewards, want_flags=want_flags,
            want_episodes=want_episodes)
        self.boards, self._spare = self._spare, self.boards
        self.t += steps
        return self.boards, flags, rewards, fstream, episodes

    def valid_moves(self, agent_semantics=False):
        """uint8 (n,) 4-bit masks, bit a = action a valid."""
        self.join()
        return ops.valid_moves(self.boards, agent_semantics)

    def valid_moves_bool(self):
        m = self.valid_moves()
        return torch.stack([(m >> a) & 1 for a in range(4)], dim=1).bool()

    def obs(self):
        """float32 (n,16): log2(tile)/15 (PPOAgent.normalize_state)."""
        self.join()
        return ops.obs(self.boards)

    def state_i32(self):
        """int32 (n,16) real tile values -- the reference's get_state() layout."""
        self.join()
        return ops.unpack(self.boards)

    def random_actions(self):
        return ops.synth_actions(self.n, self.seed, self.t, self.id_base, device=self.device)


class BatchedBeamSearch:
    """BeamSearchAgent.get_action for many games at once (reference agents/beam_search_agent.py:71-181)."""

    def __init__(self, beam_width=10, search_depth=15, seed=0x2048, early_game_threshold=512,
                 mid_game_threshold=1024, fixed_down=False):
        if not (1 <= int(beam_width) <= L.BEAM_MAX_WIDTH):
            raise ValueError("g2048: beam width must be in 1..%d" % L.BEAM_MAX_WIDTH)
        self.beam_width, self.search_depth = int(beam_width), int(search_depth)
        self.early_game_threshold, self.mid_game_threshold = int(early_game_threshold), int(mid_game_threshold)
        self.seed, self.fixed_down = int(seed), bool(fixed_down)
        self.decisions = 0
        self._history = None        # ops.BeamHistory of this agent's call sequence (the depth-balanced block order of large batches)

    def get_actions(self, boards, valid_mask=None, game_id_base=0, want_expanded=False):
        """boards uint8 (n,16). Returns (actions uint8, probs float32[, expanded])."""
        if self._history is None or ops._dev_index(self._history.device) != ops._dev_index(boards.device):
            self._history = ops.BeamHistory(boards.device)
        res = ops.beam_get_action(boards, self.beam_width, self.search_depth, valid_mask, self.early_game_threshold,
                                  self.mid_game_threshold, self.seed, self.decisions, game_id_base, self.fixed_down,
                                  want_expanded, history=self._history)
        self.decisions += 1
        return res

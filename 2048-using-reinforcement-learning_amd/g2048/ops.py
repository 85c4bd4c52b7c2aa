"""Tensor-in / tensor-out wrappers of the C-ABI (one function per entry point of include/g2048.h).

Boards are torch.uint8 tensors of shape (n, 16) holding log2 codes (0 = empty), on a ROCm device.
Every call is enqueued on torch's current stream of the boards' device and returns immediately.
"""
import torch

from . import _lib as L


def _dev(t):
    return t.device


class KeyBlock:
    """Device-resident per-move RNG keys (include/g2048.h, "graph-replayable loops"). `advance()` enqueues the
    one-thread kernel that derives the keys of move `counter` and increments the counter; the ops below accept
    `keyblock=` instead of a host step index, which makes a captured hipGraph of one move replayable."""

    def __init__(self, seed, start=0, device="cuda"):
        self.seed = int(seed)
        self.device = torch.device(device)
        self.words = torch.zeros(L.KEYBLOCK_WORDS, dtype=torch.int32, device=self.device)
        self.counter = torch.full((1,), int(start), dtype=torch.int64, device=self.device)

    def advance(self):
        L.call(self.device, L.lib().g2048_keys_advance, self.words.data_ptr(), self.counter.data_ptr(), L.u64(self.seed),
               L.stream_ptr(self.device))
        return self


def _require_scores(scores):
    """scores are 32-bit on the device (the ABI's uint32); torch code usually holds them as int32."""
    L.require_device_tensor(scores, torch.uint32 if scores.dtype == torch.uint32 else torch.int32, None, "scores")


def step(boards, actions, scores, seed, step_index, id_base=0, out=None, reward=None, flags=None,
         reward_f64=False, auto_reset=False, tune=0, keyblock=None, noop_actions=False):
    """Game2048Env.step for every board (reference environment/game_2048.py:170-210).

    scores (uint32) is updated in place. Returns (boards_out, reward, flags); flags bit0 = done,
    bit1 = valid move, bits 3..7 = max log2 code. `out` may be `boards` for an in-place step.
    actions=None: random playout, the kernel draws the uniform actions synth_actions(seed, step_index) would give.
    noop_actions: action values above 3 move nothing (an invalid move), as in the reference; default: low two bits count."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    n = boards.shape[0]
    random_actions = actions is None
    if random_actions:
        if keyblock is not None:
            raise ValueError("g2048: random actions need the scalar (seed, step_index) form")
        actions = boards        # placeholder for the length check below; the pointer passed is NULL
    else:
        L.require_device_tensor(actions, torch.uint8, None, "actions")
    _require_scores(scores)
    if actions.shape[0] != n or scores.shape[0] != n:
        raise ValueError("g2048: actions/scores length must equal the number of boards")
    dev = _dev(boards)
    if out is None:
        out = torch.empty_like(boards)
    rdt = torch.float64 if reward_f64 else torch.float32
    if reward is None:
        reward = torch.empty(n, dtype=rdt, device=dev)
    if flags is None:
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
    L.require_device_tensor(out, torch.uint8, (16,), "out")
    L.require_device_tensor(reward, rdt, None, "reward")
    L.require_device_tensor(flags, torch.uint8, None, "flags")
    opts = ((L.STEP_REWARD_F64 if reward_f64 else 0) | (L.STEP_AUTO_RESET if auto_reset else 0) |
            (L.STEP_RANDOM_ACTIONS if random_actions else 0) | (L.STEP_NOOP_ACTIONS if noop_actions else 0) | ((int(tune) & 3) << 8))
    act_ptr = None if random_actions else actions.data_ptr()
    if keyblock is not None:        # keys (and so seed / step index) come from the device key block
        L.call(dev, L.lib().g2048_step_dyn, boards.data_ptr(), act_ptr, out.data_ptr(), scores.data_ptr(),
               reward.data_ptr(), flags.data_ptr(), keyblock.words.data_ptr(), L.u64(id_base), n, opts, L.stream_ptr(dev))
    else:
        L.call(dev, L.lib().g2048_step, boards.data_ptr(), act_ptr, out.data_ptr(), scores.data_ptr(),
               reward.data_ptr(), flags.data_ptr(), L.u64(seed), L.u64(step_index), L.u64(id_base), n, opts,
               L.stream_ptr(dev))
    return out, reward, flags


class PreparedStep:
    """g2048_step with everything but the step index bound and checked ONCE: calling it costs one ctypes call (a few
    microseconds) instead of the tensor checks of `step`, so a host loop of back-to-back steps stays ahead of the GPU even
    at ~13 us per launch. Same semantics as `step(..., out=, reward=, flags=)` with explicit actions."""

    def __init__(self, boards, actions, scores, seed, id_base=0, *, out, reward, flags, auto_reset=False, tune=0):
        step(boards, actions, scores, seed, 0, id_base, out=out, reward=reward, flags=flags,
             reward_f64=reward.dtype == torch.float64, auto_reset=auto_reset, tune=tune)        # validates everything (and runs once)
        self._keep = (boards, actions, scores, out, reward, flags)
        self.device = boards.device
        opts = ((L.STEP_REWARD_F64 if reward.dtype == torch.float64 else 0) | (L.STEP_AUTO_RESET if auto_reset else 0) |
                ((int(tune) & 3) << 8))
        self._fn = L.lib().g2048_step
        self._head = (boards.data_ptr(), actions.data_ptr(), out.data_ptr(), scores.data_ptr(), reward.data_ptr(), flags.data_ptr(),
                      L.u64(seed))
        self._tail = (L.u64(id_base), boards.shape[0], opts)

    def __call__(self, step_index, stream_ptr=None):
        rc = self._fn(*self._head, int(step_index), *self._tail,
                      stream_ptr if stream_ptr is not None else torch.cuda.current_stream(self.device).cuda_stream)
        if rc != L.OK:
            L.check(rc)


class StepChains:
    """Launch form for independent sub-batches ("chains") of one batch of boards: contiguous slices of whole kernel blocks,
    chain 0 on the caller's stream, chains 1..C-1 on side streams of their own. A chain's launches are ordered only behind
    that chain's earlier launches, so one chain's launch head and drain overlap the others' arithmetic -- what a single
    launch per step cannot do. Boards are independent and every draw is keyed by the global board id, so the
    results are the single launch's bit for bit. This object only does the stream bookkeeping:

        sc = StepChains(n, 2, device)
        sc.fork()                                  # side streams wait for the current stream's work so far
        for t in range(K):
            for c, (lo, hi) in enumerate(sc.bounds):
                with torch.cuda.stream(sc.stream(c)): ops.step(boards[lo:hi], ..., id_base + lo, ...)
        sc.join()                                  # the current stream waits for every chain

    Inside a hipGraph capture the same calls become one graph with C parallel branches. No host synchronisation anywhere."""

    ALIGN = 256          # a chain owns whole blocks of the step kernel, and 16-byte aligned slices of every per-board array

    def __init__(self, n, chains, device="cuda"):
        n, chains = int(n), int(chains)
        if chains < 1:
            raise ValueError("g2048: chains must be at least 1")
        self.device = torch.device(device)
        per = -(-max(n, 1) // chains)
        per = -(-per // self.ALIGN) * self.ALIGN
        self.bounds = [(lo, min(lo + per, n)) for lo in range(0, max(n, 1), per)]       # fewer than `chains` slices for a small n
        self._side = None
        self._events = None
        self._open_on = None        # the stream the open chains were forked from; None = joined
        self._held = []             # inputs of open chains, kept alive until the join

    def __len__(self):
        return len(self.bounds)

    @property
    def is_open(self):
        return self._open_on is not None

    def keep_alive(self, *tensors):
        """Side streams use these tensors: should they be freed with chains open, the allocator waits for the side streams."""
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in self.bounds[1:]]
        for t in tensors:
            for s in self._side:
                t.record_stream(s)

    def hold(self, tensor):
        self._held.append(tensor)

    def fork(self):
        """Open the chains on the current stream (no-op if they are open on it already). Returns the current stream."""
        cur = torch.cuda.current_stream(self.device)
        if self._open_on is not None:
            if self._open_on == cur:
                return cur
            self.join()             # the caller switched streams with chains open: close them where they were opened
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in self.bounds[1:]]
        self._order(cur, self._side)
        self._open_on = cur
        return cur

    def _order(self, first, then):
        """Every stream of `then` waits for what `first` holds now -- with events this object keeps (Stream.wait_stream makes a
        new one per call: ~10 us of host time each, which a fork + join per step pays twice)."""
        if self._events is None:
            self._events = [torch.cuda.Event() for _ in range(1 + len(self.bounds))]
        if isinstance(then, list):
            ev = self._events[0]
            ev.record(first)
            for s in then:
                s.wait_event(ev)
        else:                       # `first` is a list here: `then` waits for each of them
            for k, s in enumerate(first):
                ev = self._events[1 + k]
                ev.record(s)
                then.wait_event(ev)

    def stream(self, c):
        """The stream chain c launches on (valid between fork() and join())."""
        return self._open_on if c == 0 else self._side[c - 1]

    def fence(self):
        """Open chains wait for everything queued on the current stream so far (inputs produced after the fork)."""
        if self._open_on is not None:
            self._order(torch.cuda.current_stream(self.device), self._side)

    def join(self):
        """The stream the chains were opened on waits for every chain's last launch. No-op when nothing is open."""
        if self._open_on is not None:
            self._order(self._side, self._open_on)
            self._open_on = None
            self._held.clear()


def step_many(boards, scores, seed, step_index0, steps, id_base=0, out=None, flags=None, reward_stream=None,
              flags_stream=None, episodes=None, reward_f64=False, auto_reset=False, want_rewards=False, want_flags=False,
              want_episodes=False, actions=None):
    """`steps` consecutive steps of every board in ONE launch (g2048_step_many): step t equals
    `step(boards, actions[t] or None, scores, seed, step_index0 + t, id_base, auto_reset=...)` bit for bit, but the boards stay
    in registers between the steps. actions=None: random playout (uniform actions drawn in the kernel); else a uint8 (steps, n)
    tensor of explicit actions. scores is updated in place; `out` may be `boards`. Returns (boards_out, flags_last,
    reward_stream or None, flags_stream or None, episodes or None); the streams are (steps, n) tensors, allocated when want_*
    is set."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    n, steps = boards.shape[0], int(steps)
    if steps < 1:
        raise ValueError("g2048: steps must be at least 1")
    if actions is not None:
        L.require_device_tensor(actions, torch.uint8, None, "actions")
        if tuple(actions.shape) != (steps, n):
            raise ValueError("g2048: actions must have shape (steps, n)")
    _require_scores(scores)
    if scores.shape[0] != n:
        raise ValueError("g2048: scores length must equal the number of boards")
    dev = _dev(boards)
    if out is None:
        out = torch.empty_like(boards)
    if flags is None:
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
    rdt = torch.float64 if reward_f64 else torch.float32
    if reward_stream is None and want_rewards:
        reward_stream = torch.empty((steps, n), dtype=rdt, device=dev)
    if flags_stream is None and want_flags:
        flags_stream = torch.empty((steps, n), dtype=torch.uint8, device=dev)
    if episodes is None and want_episodes:
        episodes = torch.empty(n, dtype=torch.int32, device=dev)
    L.require_device_tensor(out, torch.uint8, (16,), "out")
    L.require_device_tensor(flags, torch.uint8, None, "flags")
    for t, dt, name in ((reward_stream, rdt, "reward_stream"), (flags_stream, torch.uint8, "flags_stream")):
        if t is not None:
            L.require_device_tensor(t, dt, None, name)
            if tuple(t.shape) != (steps, n):
                raise ValueError("g2048: %s must have shape (steps, n)" % name)
    if episodes is not None:
        L.require_device_tensor(episodes, torch.int32, None, "episodes")
    opts = ((L.STEP_RANDOM_ACTIONS if actions is None else 0) | (L.STEP_REWARD_F64 if reward_f64 else 0) |
            (L.STEP_AUTO_RESET if auto_reset else 0))
    L.call(dev, L.lib().g2048_step_many, boards.data_ptr(), None if actions is None else actions.data_ptr(), out.data_ptr(),
           scores.data_ptr(),
           None if reward_stream is None else reward_stream.data_ptr(), None if flags_stream is None else flags_stream.data_ptr(),
           flags.data_ptr(), None if episodes is None else episodes.data_ptr(), L.u64(seed), L.u64(step_index0), steps,
           L.u64(id_base), n, opts, L.stream_ptr(dev))
    return out, flags, reward_stream, flags_stream, episodes


def reset(n, seed, epoch=0, id_base=0, device="cuda", boards=None, scores=None):
    """Game2048Env.reset for n boards (environment/game_2048.py:29-48). Returns (boards, scores)."""
    dev = torch.device(device) if boards is None else boards.device
    if boards is None:
        boards = torch.empty((n, 16), dtype=torch.uint8, device=dev)
    if scores is None:
        scores = torch.empty(n, dtype=torch.int32, device=dev)
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    _require_scores(scores)
    L.call(dev, L.lib().g2048_reset, boards.data_ptr(), scores.data_ptr(), L.u64(seed), L.u64(epoch), L.u64(id_base),
                                boards.shape[0], L.stream_ptr(dev))
    return boards, scores


def valid_moves(boards, agent_semantics=False, out=None):
    """4-bit masks (bit a = action a valid). Env semantics (game_2048.py:69-95) or the beam agent's own
    (_check_valid_moves, beam_search_agent.py:183-192 -- differs on DOWN)."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    if out is None:
        out = torch.empty(boards.shape[0], dtype=torch.uint8, device=boards.device)
    L.require_device_tensor(out, torch.uint8, None, "out")
    L.call(boards.device, L.lib().g2048_valid_moves, boards.data_ptr(), out.data_ptr(), boards.shape[0],
                                      L.VALID_AGENT if agent_semantics else L.VALID_ENV, L.stream_ptr(boards.device))
    return out


def evaluate(boards, kind, phase=None, out=None):
    """Board heuristics in f64: L.EVAL_FAST / EVAL_FULL (phase uint8 per board or None) /
    EVAL_PPO_HEURISTIC / EVAL_MONO_*."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    if phase is not None:
        L.require_device_tensor(phase, torch.uint8, None, "phase")
        if phase.shape[0] != boards.shape[0]:
            raise ValueError("g2048: phase length must equal the number of boards")
    if out is None:
        out = torch.empty(boards.shape[0], dtype=torch.float64, device=boards.device)
    L.require_device_tensor(out, torch.float64, None, "out")
    L.call(boards.device, L.lib().g2048_eval, boards.data_ptr(), int(kind), phase.data_ptr() if phase is not None else None,
                               out.data_ptr(), boards.shape[0], L.stream_ptr(boards.device))
    return out


def obs(boards, out=None, dtype=torch.float32):
    """PPOAgent.normalize_state for every board (agents/ppo_agent.py:184-195): float32 (n,16); dtype=torch.float16 /
    torch.bfloat16 give the same values rounded once more (nearest even) for reduced-precision policies."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    if out is not None:
        dtype = out.dtype
    if dtype in (torch.float16, torch.bfloat16):
        if out is None:
            out = torch.empty((boards.shape[0], 16), dtype=dtype, device=boards.device)
        L.require_device_tensor(out, dtype, (16,), "out")
        L.call(boards.device, L.lib().g2048_obs_16, boards.data_ptr(), out.data_ptr(), int(dtype == torch.bfloat16),
               boards.shape[0], L.stream_ptr(boards.device))
        return out
    if out is None:
        out = torch.empty((boards.shape[0], 16), dtype=torch.float32, device=boards.device)
    L.require_device_tensor(out, torch.float32, (16,), "out")
    L.call(boards.device, L.lib().g2048_obs_f32, boards.data_ptr(), out.data_ptr(), boards.shape[0], L.stream_ptr(boards.device))
    return out


def track_episodes(flags, alive, moves, valid_cnt, invalid_cnt, milestone_move, move_index, expanded=None,
                   expanded_sum=None, keyblock=None):
    """One-kernel bookkeeping of the evaluation loop (reference evaluate_beam_search.py:42-64); all tensors are
    updated in place. alive uint8 (n,), moves/valid_cnt/invalid_cnt int32 (n,), milestone_move int32 (n,8)."""
    n = flags.shape[0]
    L.require_device_tensor(flags, torch.uint8, None, "flags")
    L.require_device_tensor(alive, torch.uint8, None, "alive")
    for name, t in (("moves", moves), ("valid_cnt", valid_cnt), ("invalid_cnt", invalid_cnt)):
        L.require_device_tensor(t, torch.int32, None, name)
    L.require_device_tensor(milestone_move, torch.int32, (8,), "milestone_move")
    if expanded is not None:
        L.require_device_tensor(expanded, torch.int32, None, "expanded")
        L.require_device_tensor(expanded_sum, torch.int64, None, "expanded_sum")
    args = (flags.data_ptr(), expanded.data_ptr() if expanded is not None else None, alive.data_ptr(), moves.data_ptr(),
            valid_cnt.data_ptr(), invalid_cnt.data_ptr(), milestone_move.data_ptr(),
            expanded_sum.data_ptr() if expanded is not None else None)
    if keyblock is not None:
        L.call(flags.device, L.lib().g2048_track_episodes_dyn, *args, keyblock.words.data_ptr(), n, L.stream_ptr(flags.device))
    else:
        L.call(flags.device, L.lib().g2048_track_episodes, *args, int(move_index), n, L.stream_ptr(flags.device))


def sample_actions(probs, mask4=None, seed=0x2048, step_index=0, id_base=0, actions=None, prob=None, keyblock=None):
    """Masked categorical sampling (PPOAgent.get_action, agents/ppo_agent.py:211-221) for every env in one kernel.
    probs float32 (n,4); mask4 uint8 (n,) or None. Returns (actions uint8 (n,), prob float32 (n,))."""
    L.require_device_tensor(probs, torch.float32, (4,), "probs")
    n = probs.shape[0]
    if mask4 is not None:
        L.require_device_tensor(mask4, torch.uint8, None, "mask4")
    dev = probs.device
    if actions is None:
        actions = torch.empty(n, dtype=torch.uint8, device=dev)
    if prob is None:
        prob = torch.empty(n, dtype=torch.float32, device=dev)
    L.require_device_tensor(actions, torch.uint8, None, "actions")
    L.require_device_tensor(prob, torch.float32, None, "prob")
    mp = mask4.data_ptr() if mask4 is not None else None
    if keyblock is not None:
        L.call(dev, L.lib().g2048_sample_actions_dyn, probs.data_ptr(), mp, actions.data_ptr(), prob.data_ptr(),
               keyblock.words.data_ptr(), L.u64(id_base), n, L.stream_ptr(dev))
    else:
        L.call(dev, L.lib().g2048_sample_actions, probs.data_ptr(), mp, actions.data_ptr(), prob.data_ptr(), L.u64(seed),
               L.u64(step_index), L.u64(id_base), n, L.stream_ptr(dev))
    return actions, prob


_OBS_KIND = {torch.float32: L.OBS_F32, torch.float16: L.OBS_F16, torch.bfloat16: L.OBS_BF16}


def rollout_step(boards, probs, scores, seed, step_index, id_base=0, *, mask=None, out=None, actions=None, prob=None,
                 reward=None, flags=None, obs_next=None, mask_next=None, next_boards=None, state_maxcode=None,
                 auto_reset=True, step_counter=None):
    """One PPO rollout step of every env in ONE launch (include/g2048.h, g2048_rollout_step): masked sampling from
    `probs` (agents/ppo_agent.py:211-221) -> Game2048Env.step (environment/game_2048.py:170-210) -> next observation
    (ppo_agent.py:184-195, dtype of `obs_next`) and next valid-move mask (game_2048.py:69-95), all from registers.
    reward may be float32 or float64. step_counter: int64 device tensor added to step_index inside the kernel (for
    replayable hipGraphs). Returns (boards_out, actions, prob, reward, flags)."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    L.require_device_tensor(probs, torch.float32, (4,), "probs")
    _require_scores(scores)
    n, dev = boards.shape[0], boards.device
    if probs.shape[0] != n or scores.shape[0] != n:
        raise ValueError("g2048: probs/scores length must equal the number of boards")
    out = torch.empty_like(boards) if out is None else L.require_device_tensor(out, torch.uint8, (16,), "out")
    actions = torch.empty(n, dtype=torch.uint8, device=dev) if actions is None else L.require_device_tensor(actions, torch.uint8, None, "actions")
    prob = torch.empty(n, dtype=torch.float32, device=dev) if prob is None else L.require_device_tensor(prob, torch.float32, None, "prob")
    if reward is None:
        reward = torch.empty(n, dtype=torch.float32, device=dev)
    if reward.dtype not in (torch.float32, torch.float64):
        raise TypeError("g2048: reward must be float32 or float64")
    L.require_device_tensor(reward, reward.dtype, None, "reward")
    flags = torch.empty(n, dtype=torch.uint8, device=dev) if flags is None else L.require_device_tensor(flags, torch.uint8, None, "flags")
    opts = (L.STEP_REWARD_F64 if reward.dtype == torch.float64 else 0) | (L.STEP_AUTO_RESET if auto_reset else 0)
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    if mask is not None:
        L.require_device_tensor(mask, torch.uint8, None, "mask")
    if obs_next is not None:
        if obs_next.dtype not in _OBS_KIND:
            raise TypeError("g2048: obs_next must be float32, float16 or bfloat16")
        L.require_device_tensor(obs_next, obs_next.dtype, (16,), "obs_next")
        opts |= _OBS_KIND[obs_next.dtype] << L.ROLLOUT_OBS_SHIFT
    if mask_next is not None:
        L.require_device_tensor(mask_next, torch.uint8, None, "mask_next")
    if next_boards is not None:
        L.require_device_tensor(next_boards, torch.uint8, (16,), "next_boards")
    if state_maxcode is not None:
        L.require_device_tensor(state_maxcode, torch.uint8, None, "state_maxcode")
    if step_counter is not None:
        L.require_device_tensor(step_counter, torch.int64, None, "step_counter")
    for name, t in (("actions", actions), ("prob", prob), ("reward", reward), ("flags", flags), ("out", out), ("mask", mask),
                    ("obs_next", obs_next), ("mask_next", mask_next), ("next_boards", next_boards), ("state_maxcode", state_maxcode)):
        if t is not None and t.shape[0] != n:
            raise ValueError("g2048: %s must have one row per env" % name)
    L.call(dev, L.lib().g2048_rollout_step, boards.data_ptr(), probs.data_ptr(), ptr(mask), out.data_ptr(), scores.data_ptr(),
           actions.data_ptr(), prob.data_ptr(), reward.data_ptr(), flags.data_ptr(), ptr(obs_next), ptr(mask_next),
           ptr(next_boards), ptr(state_maxcode), L.u64(seed), L.u64(step_index), ptr(step_counter), L.u64(id_base), n, opts,
           L.stream_ptr(dev))
    return out, actions, prob, reward, flags


class SeenStates:
    """The `seen_states` set and `highest_tile_seen` of PPOAgent (agents/ppo_agent.py:171-176) for ordered batches of
    transitions, resident on the GPU: an open-addressing hash set keyed by the 16-byte board (include/g2048.h,
    g2048_seen_insert) that grows by rehashing, the running highest log2 code, and the running transition index."""

    def __init__(self, device="cuda", capacity_log2=20):
        self.device = torch.device(device)
        self.capacity_log2 = int(capacity_log2)
        self.table = torch.zeros((1 << self.capacity_log2, L.SEEN_SLOT_BYTES), dtype=torch.uint8, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.highest = torch.ones(1, dtype=torch.int32, device=self.device)     # log2 code of tile 2 (ppo_agent.py:171)
        self.index = 0              # transitions presented so far (global order)
        self._count_bound = 0       # host-side upper bound of the number of keys in the table (see reserve)

    def reserve(self, n_new):
        """Make room for n_new more transitions without overflowing: the table is kept at most half full of the keys it
        COULD hold. The host tracks an upper bound of the key count (the last count read back + every transition presented
        since); only when that bound no longer fits does it read the true count back (the one host sync, also the point where
        the overflow flag is looked at) and, if needed, grow by rehashing into a zeroed table with twice the room required --
        so in the steady state of a long run (many more keys than transitions per batch) batches are enqueued without any
        host round trip."""
        n_new = int(n_new)
        if 2 * (self._count_bound + n_new) <= (1 << self.capacity_log2):
            self._count_bound += n_new
            return
        self.assert_ok()                                    # host sync: overflow flag ...
        count = int(self.count.item())                      # ... and the true key count
        need = count + n_new
        if 2 * (need + n_new) > (1 << self.capacity_log2):      # (room for one more batch like this one under the bound, too)
            log2 = self.capacity_log2
            while (1 << log2) < 4 * need:                   # twice the room needed now: the next batches fit under the bound
                log2 += 1
            if log2 > 31:
                raise RuntimeError("g2048: seen-states table would exceed 2^31 slots")
            new = torch.zeros((1 << log2, L.SEEN_SLOT_BYTES), dtype=torch.uint8, device=self.device)
            L.call(self.device, L.lib().g2048_seen_rehash, self.table.data_ptr(), self.capacity_log2, new.data_ptr(), log2,
                   self.overflow.data_ptr(), L.stream_ptr(self.device))
            self.table, self.capacity_log2 = new, log2
        self._count_bound = need

    def assert_ok(self):
        """Raise if an insert ever found the table full or gave up probing (host sync). reserve() makes that impossible for
        batches that go through remember_shaping; call this after the LAST batch of a run, or pass check=True there."""
        flag = int(self.overflow.item())
        if flag:
            raise RuntimeError("g2048: the seen-states table overflowed (flag 0x%x): novelty terms of the batches since the last "
                               "check are not trustworthy -- it is sized by reserve(); was it bypassed?" % flag)

    def __len__(self):
        return int(self.count.item())


def remember_shaping(seen, next_boards, state_maxcode, flags, env_reward, out=None, want_novel=False, check=False):
    """PPOAgent.remember's stored reward (agents/ppo_agent.py:234-269) for an ORDERED batch of transitions, with the
    reference's sequential semantics for both stateful terms, continuing from `seen` (a SeenStates). Inputs are flat in
    order: next_boards uint8 (n,16) -- the next state BEFORE any auto-reset --, state_maxcode uint8 (n,), flags uint8 (n,)
    as the step wrote them, env_reward float64 (n,). Returns shaped float64 (n,) [, novel uint8 (n,)]. check=True reads the
    table's overflow flag back after THIS batch (one host sync) and raises if it is set; without it the flag is looked at
    the next time the table has to be re-sized, or by seen.assert_ok()."""
    L.require_device_tensor(next_boards, torch.uint8, (16,), "next_boards")
    n, dev = next_boards.shape[0], next_boards.device
    L.require_device_tensor(state_maxcode, torch.uint8, None, "state_maxcode")
    L.require_device_tensor(flags, torch.uint8, None, "flags")
    L.require_device_tensor(env_reward, torch.float64, None, "env_reward")
    if not (state_maxcode.shape[0] == flags.shape[0] == env_reward.shape[0] == n):
        raise ValueError("g2048: all transition arrays must have the same length")
    if out is None:
        out = torch.empty(n, dtype=torch.float64, device=dev)
    L.require_device_tensor(out, torch.float64, None, "out")
    if n == 0:
        return (out, torch.empty(0, dtype=torch.uint8, device=dev)) if want_novel else out

    def aligned(t, a):      # slices of larger tensors are contiguous but may start anywhere; the kernels use 16-byte loads
        return t if t.data_ptr() % a == 0 else t.clone()
    next_boards, flags, env_reward = aligned(next_boards, 16), aligned(flags, 16), aligned(env_reward, 8)
    seen.reserve(n)
    lib, st = L.lib(), L.stream_ptr(dev)
    ws = torch.empty(int(lib.g2048_shaping_scan_workspace(n)), dtype=torch.uint8, device=dev)
    prev_highest = torch.empty(n, dtype=torch.uint8, device=dev)
    slots = torch.empty(n, dtype=torch.int32, device=dev)
    novel = torch.empty(n, dtype=torch.uint8, device=dev) if want_novel else None
    L.call(dev, lib.g2048_shaping_scan, flags.data_ptr(), prev_highest.data_ptr(), seen.highest.data_ptr(), ws.data_ptr(), n, st)
    L.call(dev, lib.g2048_seen_insert, next_boards.data_ptr(), L.u64(seen.index), seen.table.data_ptr(), seen.capacity_log2,
           seen.count.data_ptr(), seen.overflow.data_ptr(), slots.data_ptr(), n, st)
    L.call(dev, lib.g2048_shaping_apply, next_boards.data_ptr(), state_maxcode.data_ptr(), flags.data_ptr(), env_reward.data_ptr(),
           prev_highest.data_ptr(), seen.table.data_ptr(), slots.data_ptr(), L.u64(seen.index), out.data_ptr(),
           novel.data_ptr() if novel is not None else None, n, st)
    seen.index += n
    if check:
        seen.assert_ok()
    return (out, novel) if want_novel else out


def simulate_move(boards, actions, highest_code=None):
    """Game2048Env.simulate_move for every (board, action) (environment/game_2048.py:341-387).
    Returns (succ uint8 (n,32,16), reward float64 (n,32), done bool (n,32), count uint8 (n,)); only the first
    count[i] slots of row i are successors, in the reference's order."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    L.require_device_tensor(actions, torch.uint8, None, "actions")
    n = boards.shape[0]
    if actions.shape[0] != n:
        raise ValueError("g2048: actions length must equal the number of boards")
    if highest_code is not None:
        L.require_device_tensor(highest_code, torch.uint8, None, "highest_code")
    dev = boards.device
    succ = torch.empty((n, 32, 16), dtype=torch.uint8, device=dev)
    reward = torch.empty((n, 32), dtype=torch.float64, device=dev)
    done = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    count = torch.empty(n, dtype=torch.uint8, device=dev)
    L.call(dev, L.lib().g2048_simulate_move, boards.data_ptr(), actions.data_ptr(),
                                        highest_code.data_ptr() if highest_code is not None else None,
                                        succ.data_ptr(), reward.data_ptr(), done.data_ptr(), count.data_ptr(), n,
                                        L.stream_ptr(dev))
    return succ, reward, done.bool(), count


def simulate_move_sampled(boards, actions, seed=0x2048, step_index=0, id_base=0):
    """The hybrid agent's simulate_move (agents/hybrid.py:578-629) for every (board, action): up to three sampled empty
    cells x {2, 4}, rewards weighted 0.9 / 0.1. Returns (succ uint8 (n,8,16), reward float64 (n,8), done bool (n,8),
    count uint8 (n,)); only the first count[i] slots of row i are successors."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    L.require_device_tensor(actions, torch.uint8, None, "actions")
    n = boards.shape[0]
    if actions.shape[0] != n:
        raise ValueError("g2048: actions length must equal the number of boards")
    dev = boards.device
    succ = torch.empty((n, 8, 16), dtype=torch.uint8, device=dev)
    reward = torch.empty((n, 8), dtype=torch.float64, device=dev)
    done = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    count = torch.empty(n, dtype=torch.uint8, device=dev)
    L.call(dev, L.lib().g2048_simulate_move_sampled, boards.data_ptr(), actions.data_ptr(), succ.data_ptr(), reward.data_ptr(),
           done.data_ptr(), count.data_ptr(), L.u64(seed), L.u64(step_index), L.u64(id_base), n, L.stream_ptr(dev))
    return succ, reward, done.bool(), count


def pack(tiles, out=None):
    """int32 (n,16) real tile values (the reference's state layout) -> packed codes."""
    L.require_device_tensor(tiles, torch.int32, (16,), "tiles")
    if out is None:
        out = torch.empty((tiles.shape[0], 16), dtype=torch.uint8, device=tiles.device)
    L.call(tiles.device, L.lib().g2048_pack_i32, tiles.data_ptr(), out.data_ptr(), tiles.shape[0], L.stream_ptr(tiles.device))
    return out


def unpack(boards, out=None):
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    if out is None:
        out = torch.empty((boards.shape[0], 16), dtype=torch.int32, device=boards.device)
    L.call(boards.device, L.lib().g2048_unpack_i32, boards.data_ptr(), out.data_ptr(), boards.shape[0], L.stream_ptr(boards.device))
    return out


def synth_boards(n, seed=0x2048, id_base=0, p_empty=0.30, max_code=11, device="cuda", out=None):
    """Synthetic boards of the benchmark configs, generated on the device."""
    if out is None:
        out = torch.empty((n, 16), dtype=torch.uint8, device=device)
    L.require_device_tensor(out, torch.uint8, (16,), "out")
    L.call(out.device, L.lib().g2048_synth_boards, out.data_ptr(), L.u64(seed), L.u64(id_base), out.shape[0],
           int(round(p_empty * 65536)), int(max_code), L.stream_ptr(out.device))
    return out


def synth_actions(n, seed=0x2048, step_index=0, id_base=0, device="cuda", out=None):
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=device)
    L.require_device_tensor(out, torch.uint8, None, "out")
    L.call(out.device, L.lib().g2048_synth_actions, out.data_ptr(), L.u64(seed), L.u64(step_index), L.u64(id_base), out.shape[0],
                                        L.stream_ptr(out.device))
    return out


def metrics(boards, scores=None, flags=None, expanded=None, out=None):
    """Per-shard metric vector (int64[24]): n, sum(score), #done, sum(expanded), hist of max code 0..17."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    if out is None:
        out = torch.zeros(24, dtype=torch.int64, device=boards.device)
    L.call(boards.device, L.lib().g2048_metrics, boards.data_ptr(), scores.data_ptr() if scores is not None else None,
                                  flags.data_ptr() if flags is not None else None,
                                  expanded.data_ptr() if expanded is not None else None,
                                  out.data_ptr(), boards.shape[0], L.stream_ptr(boards.device))
    return out


class BeamHistory:
    """The caller-owned state of the depth-balanced block order (include/g2048.h, g2048_beam_get_action_hist): the history
    buffer the blocks of one call file their games in for the next call, and the call index. One object serves ONE sequence of
    calls on one stream of one device; `BatchedBeamSearch` owns one, `beam_get_action(balanced_order=True)` keeps one per
    (device, stream) in a small registry. Thread-safe: advancing the index and enqueueing the launch happen under the object's
    lock, so two host threads that share it still hand the library consecutive indices in launch order (the device side
    degrades to caller order on any inconsistency anyway -- results never depend on the order)."""

    def __init__(self, device="cuda"):
        import threading
        self.device = torch.device(device)
        self.lock = threading.Lock()
        self.buf, self.n, self.calls = None, 0, 0

    def _prepare(self, n, need):
        """(buffer, call index) for the next call over n roots; the lock is held by the caller."""
        if self.buf is None or self.buf.numel() < need or self.n != n:
            if self.buf is None or self.buf.numel() < need:
                self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.buf.zero_()
            self.n, self.calls = n, 0
        self.calls += 1
        if self.calls >= 0x7fffffff:
            self.buf.zero_()
            self.calls = 1
        return self.buf, self.calls


class _PerStream:
    """Scratch kept per (device, stream) for the convenience path of beam_get_action: at most `cap` entries, least recently
    used evicted (a stream handle that is never used again does not pin its buffer for the life of the process)."""

    def __init__(self, cap=16):
        import collections
        import threading
        self.cap, self.lock, self.items = cap, threading.Lock(), collections.OrderedDict()

    def get(self, key, make):
        with self.lock:
            v = self.items.get(key)
            if v is None:
                v = self.items[key] = make()
                while len(self.items) > self.cap:
                    self.items.popitem(last=False)
            else:
                self.items.move_to_end(key)
            return v


class _ScratchBox:
    """One (device, stream)'s scratch buffer and the lock that makes "grow if needed + enqueue" one step."""

    def __init__(self):
        import threading
        self.buf, self.lock = None, threading.Lock()


_BEAM_WS = _PerStream()
_BEAM_HIST = _PerStream()


def _dev_index(dev):
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _size_query(dev, fn, n):
    """The library's workspace / history size queries depend on the CURRENT device's compute-unit count: ask with the tensors'
    device current, as the launch itself will run."""
    idx = _dev_index(dev)
    if torch.cuda.current_device() == idx:
        return int(fn(n))
    with torch.cuda.device(idx):
        return int(fn(n))


def beam_get_action(roots, width, depth, valid_mask=None, early_threshold=512, mid_threshold=1024,
                    seed=0x2048, step_index=0, game_id_base=0, fixed_down=False, want_expanded=False, keyblock=None,
                    rank_by_counting=False, balanced_order=True, out=None, history=None):
    """BeamSearchAgent.get_action for every root (agents/beam_search_agent.py:71-181).
    Returns (actions uint8, probs float32[, expanded int32]). balanced_order: from 4096 roots on the blocks take the games in
    a depth-balanced order (same results, shorter launch): True = the order the previous call left behind in `history` (a
    BeamHistory the caller owns; None: one kept per (device, stream), shared by whoever calls on that stream -- safe from any
    host thread; the first call of a sequence runs in caller order), "sort" = an order from this call's own roots (one more
    small launch), False = caller order."""
    L.require_device_tensor(roots, torch.uint8, (16,), "roots")
    n = roots.shape[0]
    if not (1 <= int(width) <= L.BEAM_MAX_WIDTH):
        raise ValueError("g2048: beam width must be in 1..%d" % L.BEAM_MAX_WIDTH)
    if valid_mask is not None:
        L.require_device_tensor(valid_mask, torch.uint8, None, "valid_mask")
        if valid_mask.shape[0] != n:
            raise ValueError("g2048: valid_mask length must equal the number of roots")
    dev = roots.device
    if out is not None:             # (actions, probs, expanded) buffers to reuse, e.g. inside a captured graph
        actions, probs, expanded = out
    else:
        actions = torch.empty(n, dtype=torch.uint8, device=dev)
        probs = torch.empty(n, dtype=torch.float32, device=dev)
        expanded = torch.empty(n, dtype=torch.int32, device=dev) if want_expanded else None
    head = (roots.data_ptr(), valid_mask.data_ptr() if valid_mask is not None else None, actions.data_ptr(), probs.data_ptr(),
            expanded.data_ptr() if expanded is not None else None, int(width), int(depth), int(early_threshold),
            int(mid_threshold))
    tail = (L.u64(game_id_base), n, (L.BEAM_FIXED_DOWN if fixed_down else 0) | (L.BEAM_RANK_BY_COUNTING if rank_by_counting else 0),
            L.stream_ptr(dev))
    if keyblock is not None:
        L.call(dev, L.lib().g2048_beam_get_action_dyn, *head, keyblock.words.data_ptr(), *tail)
    else:
        # the depth-balanced block order of large batches. Default: the order of the PREVIOUS call of the same sequence (its
        # blocks file their games into the history buffer, g2048_beam_get_action_hist -- no launch of its own). With
        # balanced_order="sort" the order comes from this call's own roots (beam_order_kernel, one more launch). Neither while
        # the stream is being captured: the cached tensors must not come from a graph's private pool.
        capturing = torch.cuda.is_current_stream_capturing()
        key = (_dev_index(dev), tail[-1])
        if balanced_order == "sort":
            need = 0 if capturing else _size_query(dev, L.lib().g2048_beam_workspace_bytes, n)
            if need:
                # size check, allocation and enqueue under the entry's own lock: two host threads on one stream can neither
                # replace each other's buffer before the launch nor launch into a buffer another thread just dropped (an entry
                # evicted meanwhile keeps its buffer alive through `ws` until this launch is enqueued; the caching allocator's
                # stream ordering covers the rest, the buffer having been allocated on this stream)
                box = _BEAM_WS.get(key, _ScratchBox)
                with box.lock:
                    ws = box.buf
                    if ws is None or ws.numel() < need:
                        ws = box.buf = torch.empty(need, dtype=torch.uint8, device=dev)
                    L.call(dev, L.lib().g2048_beam_get_action_ws, *head, L.u64(seed), L.u64(step_index), *tail[:-1],
                           ws.data_ptr(), need, tail[-1])
            else:
                L.call(dev, L.lib().g2048_beam_get_action_ws, *head, L.u64(seed), L.u64(step_index), *tail[:-1], None, 0, tail[-1])
        else:
            need = 0 if (not balanced_order or capturing) else _size_query(dev, L.lib().g2048_beam_history_bytes, n)
            if not need:
                L.call(dev, L.lib().g2048_beam_get_action_hist, *head, L.u64(seed), L.u64(step_index), *tail[:-1], None, 0, 0, tail[-1])
            else:
                h = history if history is not None else _BEAM_HIST.get(key, lambda: BeamHistory(dev))
                if _dev_index(h.device) != key[0]:
                    raise ValueError("g2048: this BeamHistory belongs to %s, the roots live on %s" % (h.device, dev))
                with h.lock:        # index + enqueue together: the library wants consecutive indices in launch order
                    buf, call_index = h._prepare(n, need)
                    L.call(dev, L.lib().g2048_beam_get_action_hist, *head, L.u64(seed), L.u64(step_index), *tail[:-1],
                           buf.data_ptr(), need, call_index, tail[-1])
    return (actions, probs, expanded) if (want_expanded or (out is not None and expanded is not None)) else (actions, probs)


def play_games(boards, scores, width, depth, max_moves=5000, early_threshold=512, mid_threshold=1024, seed=0x2048,
               game_id_base=0, fixed_down=False, one_phase=False, rank_by_counting=False, tuning=None, want_actions=False):
    """Every game played to completion in ONE launch (beam get_action -> env step fused per game, reference
    run_evaluation.py:48-69): one wavefront owns a game; helper wavefronts of the same launch search the boards the next
    moves can start from ahead of time, for the games that are left when the chip empties (g2048_beam.hip;
    one_phase=True plays without them -- the games are identical). boards / scores are updated in place. Returns a dict
    of per-game tensors: moves, valid_moves, invalid_moves (int32), milestone_move (int32 (n,8), -1 = never), expanded
    (int64), alive (uint8). tuning = (helpers, games_left, stuck, wait_us): the helper-wavefront parameters given explicitly
    (g2048_play_games_tuned; measurements and tests -- the games are the same for every setting). want_actions: also
    "actions", uint8 (n, max_moves): the move-set of every game, 0xFF from its end on (train.py:51,67,140-142; the input of
    `replay_games`)."""
    L.require_device_tensor(boards, torch.uint8, (16,), "boards")
    _require_scores(scores)
    n = boards.shape[0]
    if not (1 <= int(width) <= L.BEAM_MAX_WIDTH):
        raise ValueError("g2048: beam width must be in 1..%d" % L.BEAM_MAX_WIDTH)
    dev = boards.device
    out = {
        "moves": torch.zeros(n, dtype=torch.int32, device=dev), "valid_moves": torch.zeros(n, dtype=torch.int32, device=dev),
        "invalid_moves": torch.zeros(n, dtype=torch.int32, device=dev),
        "milestone_move": torch.full((n, 8), -1, dtype=torch.int32, device=dev),
        "expanded": torch.zeros(n, dtype=torch.int64, device=dev), "alive": torch.zeros(n, dtype=torch.uint8, device=dev),
    }
    if want_actions:
        out["actions"] = torch.empty((n, int(max_moves)), dtype=torch.uint8, device=dev)     # (the library fills it with 0xFF)
    # the helpers' request slots: caller-owned scratch, like every other buffer of the interface
    ws_bytes = 0 if one_phase else _size_query(dev, L.lib().g2048_play_games_workspace, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    args = (boards.data_ptr(), scores.data_ptr(), out["moves"].data_ptr(),
            out["valid_moves"].data_ptr(), out["invalid_moves"].data_ptr(), out["milestone_move"].data_ptr(),
            out["expanded"].data_ptr(), out["alive"].data_ptr(), out["actions"].data_ptr() if want_actions else None,
            int(width), int(depth), int(early_threshold),
            int(mid_threshold), int(max_moves), L.u64(seed), L.u64(game_id_base), n,
            (L.BEAM_FIXED_DOWN if fixed_down else 0) | (L.PLAY_ONE_PHASE if one_phase else 0) |
            (L.BEAM_RANK_BY_COUNTING if rank_by_counting else 0),
            ws.data_ptr() if ws is not None else None, ws_bytes)
    if tuning is not None:
        import ctypes
        t4 = (ctypes.c_uint32 * 4)(*[min(max(int(x), 0), 0xFFFFFFFF) for x in tuning])
        L.call(dev, L.lib().g2048_play_games_tuned, *args, ctypes.cast(t4, ctypes.c_void_p), L.stream_ptr(dev))
    else:
        L.call(dev, L.lib().g2048_play_games_ws, *args, L.stream_ptr(dev))
    return out


def replay_games(boards0, actions, n_moves, seed, game_ids=None, game_id_base=0, scores0=None, longest=None):
    """Recorded games replayed into their per-move histories (g2048_replay_games; reference evaluate_beam_search.py:44-50,
    :72-75: board_history / scores_history / max_tiles_history of run_game). boards0 uint8 (k,16): where each game started;
    actions uint8 (k, stride): its action bytes as `play_games(want_actions=True)` wrote them; n_moves int32 (k,); game_ids
    int64 (k,) = the GLOBAL ids the games were played under (None: game_id_base + row). Returns (boards_hist uint8
    (k, L+1, 16), score_hist int32 (k, L+1), flags_hist uint8 (k, L+1)), L = the longest game; entry t = the state before move
    t, entry n_moves = the final state; entries past a game's end are zero. longest: an upper bound of n_moves the caller
    already knows (e.g. the move cap) -- without it the histories are sized by one read-back of n_moves.max()."""
    L.require_device_tensor(boards0, torch.uint8, (16,), "boards0")
    L.require_device_tensor(actions, torch.uint8, None, "actions")
    L.require_device_tensor(n_moves, torch.int32, None, "n_moves")
    k, dev = boards0.shape[0], boards0.device
    if actions.dim() != 2 or actions.shape[0] != k or n_moves.shape[0] != k:
        raise ValueError("g2048: actions must be (k, stride) and n_moves (k,) for k start boards")
    if game_ids is not None:
        L.require_device_tensor(game_ids, torch.int64, None, "game_ids")
        if game_ids.shape[0] != k:
            raise ValueError("g2048: game_ids must have one id per game")
    if scores0 is not None:
        _require_scores(scores0)
    if longest is None:
        longest = int(n_moves.max().item()) if k else 0                          # (one host sync: the histories are sized by it)
    longest = min(int(longest), actions.shape[1])
    hist = longest + 1
    boards_hist = torch.zeros((k, hist, 16), dtype=torch.uint8, device=dev)
    score_hist = torch.zeros((k, hist), dtype=torch.int32, device=dev)
    flags_hist = torch.zeros((k, hist), dtype=torch.uint8, device=dev)
    L.call(dev, L.lib().g2048_replay_games, boards0.data_ptr(), scores0.data_ptr() if scores0 is not None else None,
           game_ids.data_ptr() if game_ids is not None else None, L.u64(game_id_base), actions.data_ptr(), actions.shape[1],
           n_moves.data_ptr(), boards_hist.data_ptr(), score_hist.data_ptr(), flags_hist.data_ptr(), hist, L.u64(seed), k,
           L.stream_ptr(dev))
    return boards_hist, score_hist, flags_hist


def env_step(board, score, record, seed, index, board_id=0, action=0, op=None):
    """One iteration of ONE env in one launch (g2048_env_step): op ENV_OP_STEP (default) / ENV_OP_RESET / ENV_OP_PEEK. board uint8
    (1,16) and score int32 (1,) are updated in place; `record` -- uint8 (80,), device memory or pinned host memory -- receives
    the state as int32 tiles, the score, the flags, the next valid-move mask and the f64 reward (include/g2048.h)."""
    L.require_device_tensor(board, torch.uint8, (16,), "board")
    _require_scores(score)
    if not isinstance(record, torch.Tensor) or record.dtype != torch.uint8 or record.numel() < L.ENV_RECORD_BYTES or not record.is_contiguous():
        raise TypeError("g2048: record must be a contiguous uint8 tensor of at least %d bytes" % L.ENV_RECORD_BYTES)
    if not (record.is_cuda or record.is_pinned()):
        raise RuntimeError("g2048: record must live in device memory or in pinned host memory")
    dev = board.device
    L.call(dev, L.lib().g2048_env_step, board.data_ptr(), score.data_ptr(), int(action) & 0xFFFFFFFF,
           L.ENV_OP_STEP if op is None else int(op), record.data_ptr(), L.u64(seed), L.u64(index), L.u64(board_id), L.stream_ptr(dev))


def minibatch_gather(obs, actions, log_probs, rewards, next_boards, flags, batch, seed, sample_index, want_indices=False, out=None):
    """PPOMemory.sample + the head of PPOAgent.update (agents/ppo_agent.py:21-50, :342-354) on a device-resident trajectory
    (g2048_minibatch_gather): `batch` distinct transitions drawn and gathered by one launch, no host sync. Inputs are flat over
    the transitions: obs (n,16) float32 / float16 / bfloat16, actions uint8, log_probs float32, rewards float32 or float64,
    next_boards uint8 (n,16) (before any auto-reset), flags uint8. Returns dict(states, actions, old_log_probs, rewards,
    next_states, dones[, indices]). out: the dict a previous call of the same batch size returned, to be overwritten (at the
    reference's batch sizes a call is mostly the host time of allocating six small tensors)."""
    if obs.dtype not in _OBS_KIND:
        raise TypeError("g2048: obs must be float32, float16 or bfloat16")
    L.require_device_tensor(obs, obs.dtype, (16,), "obs")
    n, dev = obs.shape[0], obs.device
    L.require_device_tensor(actions, torch.uint8, None, "actions")
    L.require_device_tensor(log_probs, torch.float32, None, "log_probs")
    if rewards.dtype not in (torch.float32, torch.float64):
        raise TypeError("g2048: rewards must be float32 or float64")
    L.require_device_tensor(rewards, rewards.dtype, None, "rewards")
    L.require_device_tensor(next_boards, torch.uint8, (16,), "next_boards")
    L.require_device_tensor(flags, torch.uint8, None, "flags")
    if not (actions.shape[0] == log_probs.shape[0] == rewards.shape[0] == next_boards.shape[0] == flags.shape[0] == n):
        raise ValueError("g2048: all trajectory arrays must have one entry per transition")
    batch = min(int(batch), n)                      # PPOMemory.sample: a batch larger than the buffer is the whole buffer
    if out is not None:
        if out["actions"].shape[0] != batch or out["states"].device != dev or (want_indices and "indices" not in out):
            raise ValueError("g2048: `out` must come from a call with the same batch size, device and want_indices")
        want_indices = "indices" in out
    else:
        out = {"states": torch.empty((batch, 16), dtype=torch.float32, device=dev),
               "actions": torch.empty(batch, dtype=torch.int64, device=dev),
               "old_log_probs": torch.empty(batch, dtype=torch.float32, device=dev),
               "rewards": torch.empty(batch, dtype=torch.float32, device=dev),
               "next_states": torch.empty((batch, 16), dtype=torch.float32, device=dev),
               "dones": torch.empty(batch, dtype=torch.float32, device=dev)}
        if want_indices:
            out["indices"] = torch.empty(batch, dtype=torch.int64, device=dev)
    L.call(dev, L.lib().g2048_minibatch_gather, obs.data_ptr(), _OBS_KIND[obs.dtype], actions.data_ptr(), log_probs.data_ptr(),
           rewards.data_ptr(), int(rewards.dtype == torch.float64), next_boards.data_ptr(), flags.data_ptr(), n, batch, L.u64(seed),
           L.u64(sample_index), out["states"].data_ptr(), out["actions"].data_ptr(), out["old_log_probs"].data_ptr(),
           out["rewards"].data_ptr(), out["next_states"].data_ptr(), out["dones"].data_ptr(),
           out["indices"].data_ptr() if want_indices else None, L.stream_ptr(dev))
    return out


def launch_plan(compute_units=0, resident_blocks_per_cu=0, n_games=0):
    """The chip-size arithmetic of the library (g2048_launch_plan; a host function, usable without a GPU when compute_units
    is given): dict(order_row, order_min_games, helper_cap, default_helpers)."""
    import ctypes
    out = (ctypes.c_uint32 * 4)()
    L.check(L.lib().g2048_launch_plan(int(compute_units), int(resident_blocks_per_cu), int(n_games), ctypes.cast(out, ctypes.c_void_p)))
    return dict(order_row=out[0], order_min_games=out[1], helper_cap=out[2], default_helpers=out[3])


def device_plan(width=20, n_games=4096, device="cuda"):
    """What the library asks the current device before a beam / evaluation launch (g2048_device_plan): dict(compute_units,
    play_resident_blocks_per_cu, helper_cap, default_helpers, beam_resident_blocks, beam_issue_priority)."""
    import ctypes
    out = (ctypes.c_uint32 * 6)()
    with torch.cuda.device(torch.device(device)):
        L.check(L.lib().g2048_device_plan(int(width), int(n_games), ctypes.cast(out, ctypes.c_void_p)))
    return dict(compute_units=out[0], play_resident_blocks_per_cu=out[1], helper_cap=out[2], default_helpers=out[3],
                beam_resident_blocks=out[4], beam_issue_priority=bool(out[5]))


def selftest(device="cuda"):
    r = torch.full((1,), 0xFFFF, dtype=torch.int32, device=device)
    L.call(r.device, L.lib().g2048_selftest, r.data_ptr(), L.stream_ptr(r.device))
    return int(r.item())

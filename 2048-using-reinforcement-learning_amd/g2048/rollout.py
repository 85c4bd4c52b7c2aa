"""On-device PPO rollout collector (BASELINE config 4: 65,536 envs x 128 steps).

Replaces the host-side pieces of the reference's PPO data path that sit between the env and the learner:
the `PPOMemory` deque and its per-sample Python loops (agents/ppo_agent.py:14-59), the per-step
`normalize_state` + host->device copy (`:184-202`) and the masked sampling (`:211-221`). Trajectories live
in HBM as (T, N, ...) tensors; nothing crosses PCIe during collection. The policy is any torch module
mapping float32 (N, 16) observations to action probabilities (N, 4) -- or (probs, value) like the
reference's `TransformerModel` (models/transformer.py) -- and stays stock PyTorch-ROCm: it is the consumer.
"""
import warnings

import torch

from . import _lib as L
from . import ops
from .vec import VecGame2048


def masked_sample(probs, mask4, generator=None):
    """PPOAgent.get_action's masked sampling for a batch (agents/ppo_agent.py:211-221):
    logits = log(probs + 1e-10) + (-inf where the move is invalid); a ~ Categorical(logits).
    mask4: uint8 (N,) bit a = action a valid; a row with no valid move is sampled unmasked.
    Returns (actions uint8 (N,), log_prob float32 (N,))."""
    bits = torch.stack([(mask4 >> a) & 1 for a in range(4)], dim=1).bool()
    bits = bits | ~bits.any(dim=1, keepdim=True)
    logits = torch.log(probs.float() + 1e-10).masked_fill(~bits, float("-inf"))
    logp_all = torch.log_softmax(logits, dim=1)
    gumbel = -torch.log(-torch.log(torch.rand(logits.shape, device=logits.device, generator=generator).clamp_(1e-20, 1.0)))
    actions = torch.argmax(torch.where(bits, logp_all + gumbel, torch.full_like(logp_all, float("-inf"))), dim=1)
    return actions.to(torch.uint8), logp_all.gather(1, actions[:, None]).squeeze(1)


class RolloutCollector:
    """(T, N) trajectory buffers in HBM filled by ONE g2048 launch per env step.

    Per step: policy(obs[t]) (stock torch) -> `g2048_rollout_step`, which samples the action under the valid-move mask,
    steps the env (auto-reset on) and writes action, probability, reward, flags AND the next step's observation and
    valid-move mask straight into the trajectory from the board it holds in registers -- obs[t+1] / valid_mask[t+1] are
    outputs of step t, nothing is re-read. With use_graph (default) the whole T-step loop, policy included, is captured
    once into a hipGraph and replayed per collect(): the kernels read the step number from a device counter, so the replay
    needs no host-side arguments. shaping=True also produces PPOAgent.remember's shaped reward (agents/ppo_agent.py:234-269)
    for the trajectory in (step, env) order, stateful terms included (`ops.remember_shaping`).

    sampler="torch" keeps the unfused reference-style path (torch masked_sample + one launch per piece)."""

    def __init__(self, n_envs, n_steps, policy, device="cuda", seed=0x2048, id_base=0, shaping=False,
                 generator=None, sampler="fused", obs_dtype=torch.float32, use_graph=True, seen_capacity_log2=20,
                 minibatches=False):
        self.n, self.T = int(n_envs), int(n_steps)
        self.device = torch.device(device)
        self.policy = policy
        self.shaping = bool(shaping)
        self.generator = generator
        if sampler not in ("fused", "torch"):
            raise ValueError("sampler must be 'fused' (g2048_rollout_step, counter RNG) or 'torch' (masked_sample)")
        if shaping and sampler != "fused":
            raise ValueError("shaping=True needs sampler='fused' (the fused step records what remember() consumes)")
        if minibatches and sampler != "fused":
            raise ValueError("minibatches=True needs sampler='fused' (the fused step records the next states sample() gathers)")
        self.minibatches = bool(minibatches) or self.shaping     # sample() needs the next states before auto-reset, as shaping does
        self._samples = 0           # sample() calls so far (RNG index of the minibatch permutation)
        self._seed = int(seed)
        self._filled = False        # a collect() has filled the trajectory buffers
        self._collects_since_check = 0
        self.sampler = sampler
        self.env = VecGame2048(self.n, device=self.device, seed=seed, id_base=id_base, auto_reset=True)
        d, T, n = self.device, self.T, self.n
        # one extra row: row T of obs / masks is what the last step writes = the state the next collect() starts from
        self._obs = torch.empty((T + 1, n, 16), dtype=obs_dtype, device=d)     # float32 (reference) or float16 / bfloat16
        self._masks = torch.empty((T + 1, n), dtype=torch.uint8, device=d)
        self.obs, self.masks, self.last_obs = self._obs[:T], self._masks[:T], self._obs[T]
        self.actions = torch.empty((T, n), dtype=torch.uint8, device=d)
        self.logp = torch.empty((T, n), dtype=torch.float32, device=d)
        self.values = torch.zeros((T, n), dtype=torch.float32, device=d)
        self.flags = torch.empty((T, n), dtype=torch.uint8, device=d)
        if self.shaping:
            # remember() needs the f64 env reward, the next state before auto-reset and the max tile of the state acted in
            self.rewards64 = torch.empty((T, n), dtype=torch.float64, device=d)
            self.rewards = torch.empty((T, n), dtype=torch.float32, device=d)
            self.next_boards = torch.empty((T, n, 16), dtype=torch.uint8, device=d)
            self.state_maxcode = torch.empty((T, n), dtype=torch.uint8, device=d)
            self.shaped = torch.empty((T, n), dtype=torch.float64, device=d)
            self.seen = ops.SeenStates(d, seen_capacity_log2)
        else:
            self.rewards64 = self.next_boards = self.state_maxcode = self.shaped = self.seen = None
            self.rewards = torch.empty((T, n), dtype=torch.float32, device=d)
            if self.minibatches:
                self.next_boards = torch.empty((T, n, 16), dtype=torch.uint8, device=d)
        self.env_steps = 0
        self.use_graph = bool(use_graph) and sampler == "fused"
        self._graph = None
        self._counter = torch.zeros(1, dtype=torch.int64, device=d)     # device copy of env.t at the start of a collect
        self._primed = False

    # -- one env step, fused (1 g2048 launch) or reference-style (obs / mask / sample / step launches) --------------------
    def _policy(self, t):
        out = self.policy(self._obs[t])
        probs, value = out if isinstance(out, (tuple, list)) else (out, None)
        if value is not None:
            self.values[t].copy_(value.reshape(-1))
        return probs.float().contiguous()

    def _step_fused(self, t, counter):
        env = self.env
        probs = self._policy(t)
        ops.rollout_step(env.boards, probs, env.scores, env.seed, t if counter is not None else env.t, env.id_base,
                         mask=self._masks[t], out=env._spare, actions=self.actions[t], prob=self.logp[t],
                         reward=self.rewards64[t] if self.shaping else self.rewards[t], flags=self.flags[t],
                         obs_next=self._obs[t + 1], mask_next=self._masks[t + 1],
                         next_boards=self.next_boards[t] if self.next_boards is not None else None,
                         state_maxcode=self.state_maxcode[t] if self.shaping else None, auto_reset=True,
                         step_counter=counter)
        env.boards, env._spare = env._spare, env.boards

    def _step_torch(self, t):
        env = self.env
        probs = self._policy(t)
        a, lp = masked_sample(probs, self._masks[t], self.generator)
        self.actions[t].copy_(a)
        self.logp[t].copy_(lp)
        ops.step(env.boards, self.actions[t], env.scores, env.seed, env.t, env.id_base, out=env._spare,
                 reward=self.rewards[t], flags=self.flags[t], auto_reset=True)
        env.boards, env._spare = env._spare, env.boards
        ops.obs(env.boards, out=self._obs[t + 1])
        ops.valid_moves(env.boards, out=self._masks[t + 1])

    def _prime(self):
        """Row 0 of obs / masks for the very first collect (afterwards the last step of a collect provides it)."""
        ops.obs(self.env.boards, out=self._obs[0])
        ops.valid_moves(self.env.boards, out=self._masks[0])
        self._primed = True

    def _loop_fused(self, counter):
        for t in range(self.T):
            self._step_fused(t, counter)
        if self.T % 2:          # an odd number of buffer swaps: copy back so that a replay starts from the same buffers
            self.env._spare.copy_(self.env.boards)
            self.env.boards, self.env._spare = self.env._spare, self.env.boards

    def _capture(self):
        """Capture the T-step loop (policy included) once. A warm-up pass really runs first (lazy initialisation inside
        the policy must not happen under capture); its effects are undone by restoring the env state."""
        env, dev = self.env, self.device
        tracked = (env.boards, env._spare, env.scores, self._obs[0], self._masks[0])
        saved = [x.clone() for x in tracked]
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        graph = None
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                self._loop_fused(self._counter)
                side.synchronize()
                for dst, src in zip(tracked, saved):
                    dst.copy_(src)
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    self._loop_fused(self._counter)
                    self._counter.add_(self.T)
        except Exception as exc:                # noqa: BLE001 -- e.g. a policy that cannot be captured
            warnings.warn("g2048.RolloutCollector: hipGraph capture of the rollout failed (%s: %s); collecting with "
                          "one launch sequence per step" % (type(exc).__name__, exc), RuntimeWarning, stacklevel=3)
            graph = None
            torch.cuda.synchronize(dev)
        cur.wait_stream(side)
        for dst, src in zip(tracked, saved):
            dst.copy_(src)
        self._counter.fill_(env.t)
        self._graph = graph
        if graph is None:
            self.use_graph = False

    @torch.no_grad()
    def collect(self):
        """T steps of every env. Returns the buffers (views, overwritten by the next collect)."""
        env = self.env
        if not self._primed:
            self._prime()
        else:                   # the state the previous collect ended in is row T of obs / masks
            self._obs[0].copy_(self._obs[self.T])
            self._masks[0].copy_(self._masks[self.T])
        if self.sampler == "torch":
            for t in range(self.T):
                self._step_torch(t)
                env.t += 1
        else:
            if self.use_graph and self._graph is None:
                self._capture()
            if self._graph is not None:
                self._graph.replay()            # the device counter advances by T inside the graph
            else:
                base = env.t
                for t in range(self.T):
                    env.t = base + t
                    self._step_fused(t, None)
                env.t = base
            env.t += self.T
            torch.log_(self.logp)               # the kernel stores the probability; the reference keeps its log
        if self.shaping:
            self.rewards.copy_(self.rewards64)
            T, n = self.T, self.n
            ops.remember_shaping(self.seen, self.next_boards.view(T * n, 16), self.state_maxcode.view(T * n),
                                 self.flags.view(T * n), self.rewards64.view(T * n), out=self.shaped.view(T * n))
            self._collects_since_check += 1
            if self._collects_since_check >= self.CHECK_EVERY:     # the table's overflow flag: one host sync every so often
                self.check()
        self._filled = True
        self.env_steps += self.T * self.n
        return {
            "obs": self.obs, "valid_mask": self.masks, "actions": self.actions, "log_prob": self.logp,
            "values": self.values, "rewards": self.rewards,
            "dones": (self.flags & L.FLAG_DONE).bool(), "valid_move": (self.flags & L.FLAG_VALID).bool(),
            "max_code": self.flags >> L.FLAG_MAXCODE_SHIFT, "shaping": self.shaped, "last_obs": self.last_obs,
            "last_valid_mask": self._masks[self.T],
        }

    CHECK_EVERY = 16        # collect() calls between two looks at the seen-states table's overflow flag (shaping=True)

    def check(self):
        """Raise if the seen-states table of the shaping terms ever overflowed (ops.SeenStates.assert_ok: one host sync).
        collect() calls it every CHECK_EVERY collects; call it yourself after the last collect of a run."""
        self._collects_since_check = 0
        if self.seen is not None:
            self.seen.assert_ok()

    def sample(self, batch_size, generator=None, want_indices=False, out=None):
        """PPOMemory.sample(batch_size) (agents/ppo_agent.py:21-50) over the transitions of the last collect(), as the tensors
        PPOAgent.update builds from it (:342-354): dict(states float32 (B,16) normalized, actions int64 (B,), old_log_probs
        float32 (B,), rewards float32 (B,) -- the shaped reward remember() stores when shaping=True, else the env's --,
        next_states float32 (B,16) normalized (the state the env returned, before any auto-reset), dones float32 (B,)).
        B distinct transitions (without replacement; a batch larger than the buffer is the whole buffer), drawn and gathered
        by ONE launch (g2048_minibatch_gather), no host synchronisation. The draw is keyed by (the collector's seed, the number
        of sample() calls so far); generator: a CPU torch.Generator to take the key from instead; out: the dict a previous call
        of the same batch size returned, to be overwritten instead of allocating six tensors again."""
        if not self.minibatches:
            raise RuntimeError("RolloutCollector.sample needs minibatches=True (or shaping=True): the next states are not recorded")
        if not self._filled:
            raise RuntimeError("RolloutCollector.sample: collect() first")
        seed = self._seed
        if generator is not None:
            if generator.device.type != "cpu":
                raise ValueError("sample(generator=...): a CPU generator (its draw must not synchronise the device)")
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        T, n = self.T, self.n
        rewards = self.shaped if self.shaping else self.rewards
        out = ops.minibatch_gather(self.obs.reshape(T * n, 16), self.actions.view(T * n), self.logp.view(T * n),
                                   rewards.view(T * n), self.next_boards.view(T * n, 16), self.flags.view(T * n),
                                   batch_size, seed, self._samples, want_indices=want_indices, out=out)
        self._samples += 1
        return out

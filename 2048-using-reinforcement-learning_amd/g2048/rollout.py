"""On-device PPO rollout collector (BASELINE config 4: 65,536 envs x 128 steps).

Replaces the host-side pieces of the reference's PPO data path that sit between the env and the learner:
the `PPOMemory` deque and its per-sample Python loops (agents/ppo_agent.py:14-59), the per-step
`normalize_state` + host->device copy (`:184-202`) and the masked sampling (`:211-221`). Trajectories live
in HBM as (T, N, ...) tensors; nothing crosses PCIe during collection. The policy is any torch module
mapping float32 (N, 16) observations to action probabilities (N, 4) -- or (probs, value) like the
reference's `TransformerModel` (models/transformer.py) -- and stays stock PyTorch-ROCm: it is the consumer.
"""
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
    def __init__(self, n_envs, n_steps, policy, device="cuda", seed=0x2048, id_base=0, shaping=False,
                 generator=None, sampler="fused", obs_dtype=torch.float32):
        self.n, self.T = int(n_envs), int(n_steps)
        self.device = torch.device(device)
        self.policy = policy
        self.shaping = bool(shaping)
        self.generator = generator
        if sampler not in ("fused", "torch"):
            raise ValueError("sampler must be 'fused' (g2048_sample_actions, counter RNG) or 'torch' (masked_sample)")
        self.sampler = sampler
        self.env = VecGame2048(self.n, device=self.device, seed=seed, id_base=id_base, auto_reset=True)
        d, T, n = self.device, self.T, self.n
        self.obs = torch.empty((T, n, 16), dtype=obs_dtype, device=d)      # float32 (reference) or float16 / bfloat16
        self.masks = torch.empty((T, n), dtype=torch.uint8, device=d)
        self.actions = torch.empty((T, n), dtype=torch.uint8, device=d)
        self.logp = torch.empty((T, n), dtype=torch.float32, device=d)
        self.values = torch.zeros((T, n), dtype=torch.float32, device=d)
        self.rewards = torch.empty((T, n), dtype=torch.float32, device=d)
        self.flags = torch.empty((T, n), dtype=torch.uint8, device=d)
        self.shaped = torch.empty((T, n), dtype=torch.float64, device=d) if shaping else None
        self.last_obs = torch.empty((n, 16), dtype=obs_dtype, device=d)
        self.env_steps = 0

    @torch.no_grad()
    def collect(self):
        """T steps of every env. Returns the buffers (views, overwritten by the next collect)."""
        env = self.env
        for t in range(self.T):
            ops.obs(env.boards, out=self.obs[t])
            ops.valid_moves(env.boards, out=self.masks[t])
            out = self.policy(self.obs[t])
            probs, value = out if isinstance(out, (tuple, list)) else (out, None)
            if self.sampler == "fused":      # one kernel: draw keyed by (seed, POLICY, step, global env id)
                ops.sample_actions(probs.float().contiguous(), self.masks[t], env.seed, env.t, env.id_base,
                                   actions=self.actions[t], prob=self.logp[t])
                torch.log_(self.logp[t])
            else:
                a, lp = masked_sample(probs, self.masks[t], self.generator)
                self.actions[t].copy_(a)
                self.logp[t].copy_(lp)
            if value is not None:
                self.values[t].copy_(value.reshape(-1))
            # step in place on the env's buffers, reward / flags written straight into the trajectory
            ops.step(env.boards, self.actions[t], env.scores, env.seed, env.t, env.id_base, out=env._spare,
                     reward=self.rewards[t], flags=self.flags[t], auto_reset=True)
            env.boards, env._spare = env._spare, env.boards
            env.t += 1
            if self.shaping:
                # pure terms of PPOAgent.remember on the next state (agents/ppo_agent.py:253-266); NB on a
                # finished env the "next state" stored here is the fresh board (auto-reset), as the flags say
                ops.evaluate(env.boards, L.EVAL_PPO_SHAPING, out=self.shaped[t])
        ops.obs(env.boards, out=self.last_obs)
        self.env_steps += self.T * self.n
        return {
            "obs": self.obs, "valid_mask": self.masks, "actions": self.actions, "log_prob": self.logp,
            "values": self.values, "rewards": self.rewards,
            "dones": (self.flags & L.FLAG_DONE).bool(), "valid_move": (self.flags & L.FLAG_VALID).bool(),
            "max_code": self.flags >> L.FLAG_MAXCODE_SHIFT, "shaping": self.shaped, "last_obs": self.last_obs,
        }

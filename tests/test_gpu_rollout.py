"""f2 / BASELINE config 4: PPO rollout of 65,536 parallel envs x 128 steps with a transformer policy on
PyTorch-ROCm. The env side of the trajectory (boards via observations, rewards, done flags, auto-resets) is
replayed through the oracle with the recorded actions and must match exactly."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class TinyTransformerPolicy(nn.Module):
    """Same shape as the reference's (unused) models/transformer.py: 16 tokens, d_model 64, 4 heads, 2 layers,
    actor + critic heads -- stock torch modules, random init; it is only the consumer of the rollout."""

    def __init__(self, d_model=64, nhead=4, num_layers=2):
        super().__init__()
        self.embedding = nn.Linear(1, d_model)
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=128, batch_first=True)
        self.encoder = nn.TransformerEncoder(layer, num_layers=num_layers)
        self.fc = nn.Sequential(nn.Linear(d_model * 16, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU())
        self.actor, self.critic = nn.Linear(64, 4), nn.Linear(64, 1)

    def forward(self, x):
        h = self.encoder(self.embedding(x.view(x.shape[0], 16, 1)))
        h = self.fc(h.reshape(x.shape[0], -1))
        return torch.softmax(self.actor(h), dim=-1), self.critic(h)


@pytest.fixture(scope="module")
def g2048():
    import __graft_entry__ as ge
    ge.ensure_built()
    return ge.import_package()


def replay_and_check(oracle, res, n, T, seed, id_base=0):
    b, sc = oracle.reset_batch(n, seed=seed, epoch=0, id_base=id_base)
    obs = res["obs"].cpu().numpy(); acts = res["actions"].cpu().numpy()
    rew = res["rewards"].cpu().numpy(); dones = res["dones"].cpu().numpy(); masks = res["valid_mask"].cpu().numpy()
    for t in range(T):
        assert np.array_equal(oracle.obs_batch(b), obs[t]), t
        assert np.array_equal(oracle.valid_moves_batch(b, False), masks[t]), t
        assert bool(((masks[t] >> acts[t]) & 1).all()), "sampled an invalid action"
        b, sc, r, fl = oracle.step_batch(b, acts[t], sc, seed=seed, step_index=t, id_base=id_base, opts=1)
        assert np.array_equal(r.astype(np.float32), rew[t]), t
        assert np.array_equal((fl & 1).astype(bool), dones[t]), t
    assert np.array_equal(oracle.obs_batch(b), res["last_obs"].cpu().numpy())
    return b


def test_masked_sample_semantics(g2048):
    torch.manual_seed(0)
    probs = torch.tensor([[0.7, 0.1, 0.1, 0.1]] * 40000, device=DEV)
    mask = torch.full((40000,), 0b1010, dtype=torch.uint8, device=DEV)
    a, lp = g2048.masked_sample(probs, mask)
    assert set(a.unique().tolist()) == {1, 3}
    assert abs(float((a == 1).float().mean()) - 0.5) < 0.02
    assert torch.allclose(lp, torch.full_like(lp, float(np.log(0.5))), atol=1e-5)
    a2, _ = g2048.masked_sample(probs, torch.zeros(40000, dtype=torch.uint8, device=DEV))   # no valid move: unmasked
    assert abs(float((a2 == 0).float().mean()) - 0.7) < 0.02


def test_rollout_small_with_shaping(g2048, oracle):
    torch.manual_seed(1)
    n, T = 2048, 96
    pol = TinyTransformerPolicy().to(DEV).eval()
    rc = g2048.RolloutCollector(n, T, pol, device=DEV, seed=11, id_base=5, shaping=True)
    res = rc.collect()
    b = replay_and_check(oracle, res, n, T, 11, 5)
    assert int(res["dones"].sum()) >= 0 and res["values"].abs().sum() > 0
    # shaping term of the last step's next state
    assert np.array_equal(res["shaping"][T - 1].cpu().numpy(), oracle.eval_batch(b, oracle.EVAL_PPO_SHAPING))
    res2 = rc.collect()      # a second rollout continues the same episodes (step counter keeps running)
    assert rc.env.t == 2 * T and rc.env_steps == 2 * n * T


def test_config4_rollout_65536x128(g2048, oracle):
    torch.manual_seed(2)
    n, T = 65536, 128
    pol = TinyTransformerPolicy().to(DEV).eval()
    rc = g2048.RolloutCollector(n, T, pol, device=DEV, seed=0x2048)
    torch.cuda.synchronize(); import time; t0 = time.perf_counter()
    res = rc.collect()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("config 4: %d env-steps in %.3f s = %.3e env-steps/s end-to-end (transformer policy)" % (n * T, dt, n * T / dt))
    replay_and_check(oracle, res, n, T, 0x2048)
    assert int(res["dones"].sum()) > 0          # random-ish policy: episodes end inside 128 steps and auto-reset


def test_fused_sampler_vs_oracle(g2048, oracle):
    from g2048 import ops
    torch.manual_seed(4)
    n = 300000
    probs = torch.softmax(torch.randn(n, 4, device=DEV) * 2, dim=1)
    probs[:64] = 0
    mask = torch.randint(0, 16, (n,), dtype=torch.uint8, device=DEV)
    a, pa = ops.sample_actions(probs, mask, seed=21, step_index=6, id_base=1000)
    oa, op = oracle.sample_batch(probs.cpu().numpy(), mask.cpu().numpy(), seed=21, step_index=6, id_base=1000)
    assert np.array_equal(a.cpu().numpy(), oa)
    assert np.array_equal(pa.cpu().numpy().view(np.uint32), op.view(np.uint32))          # f32 bit-exact
    a2, _ = ops.sample_actions(probs, None, seed=21, step_index=6, id_base=1000)
    oa2, _ = oracle.sample_batch(probs.cpu().numpy(), None, seed=21, step_index=6, id_base=1000)
    assert np.array_equal(a2.cpu().numpy(), oa2)


def test_rollout_torch_sampler_still_available(g2048, oracle):
    torch.manual_seed(5)
    pol = TinyTransformerPolicy().to(DEV).eval()
    rc = g2048.RolloutCollector(1024, 16, pol, device=DEV, seed=3, sampler="torch")
    replay_and_check(oracle, rc.collect(), 1024, 16, 3)

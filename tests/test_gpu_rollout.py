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


def oracle_remember_of_rollout(oracle, R, res, rc, n, T, seed, id_base, b, sc, t0):
    """remember() applied sequentially, in (step, env) order, to the transitions the collector recorded: the oracle replays
    the env with the recorded actions -- once without auto-reset for the next state remember() sees, once with it for the
    state the rollout continues from."""
    acts = res["actions"].cpu().numpy()
    shaped = np.empty((T, n), np.float64)
    for t in range(T):
        nb, _, r, fl = oracle.step_batch(b, acts[t], sc.copy(), seed=seed, step_index=t0 + t, id_base=id_base, opts=0)
        shaped[t], _ = R.batch(b, nb, r)
        b, sc, _, _ = oracle.step_batch(b, acts[t], sc, seed=seed, step_index=t0 + t, id_base=id_base, opts=1)
    return shaped, b, sc


def test_rollout_small_with_shaping(g2048, oracle):
    """RolloutCollector(shaping=True): the env side against the oracle, and the shaped reward -- PPOAgent.remember with
    its two stateful terms -- against the oracle's SEQUENTIAL remember over the same transitions, across two collects."""
    torch.manual_seed(1)
    n, T = 2048, 96
    pol = TinyTransformerPolicy().to(DEV).eval()
    rc = g2048.RolloutCollector(n, T, pol, device=DEV, seed=11, id_base=5, shaping=True, seen_capacity_log2=8)
    res = rc.collect()
    replay_and_check(oracle, res, n, T, 11, 5)
    assert res["values"].abs().sum() > 0
    R = oracle.Remember()
    b, sc = oracle.reset_batch(n, seed=11, epoch=0, id_base=5)
    want, b, sc = oracle_remember_of_rollout(oracle, R, res, rc, n, T, 11, 5, b, sc, 0)
    assert np.array_equal(res["shaping"].cpu().numpy(), want)
    res2 = rc.collect()      # a second rollout continues the same episodes, the same seen-set and highest tile
    assert rc.env.t == 2 * T and rc.env_steps == 2 * n * T
    want2, b, sc = oracle_remember_of_rollout(oracle, R, res2, rc, n, T, 11, 5, b, sc, T)
    assert np.array_equal(res2["shaping"].cpu().numpy(), want2)
    assert len(rc.seen) == R.n_seen and (1 << int(rc.seen.highest.item())) == R.highest_tile_seen
    assert rc.seen.capacity_log2 > 8          # the table grew by rehashing on the way


def test_remember_shaping_golden(g2048, oracle):
    """ops.remember_shaping against what the reference's own remember() stored (tests/golden/remember.npz), f64 ==, fed
    in pieces through a table that starts tiny (2^4 slots) so that it is rehashed several times."""
    from conftest import load_golden
    from g2048 import ops
    g = load_golden("remember.npz")
    n = g["state"].shape[0]
    nxt = torch.from_numpy(g["next_state"]).to(DEV)
    state_max = torch.from_numpy(g["state"].max(axis=1).astype(np.uint8)).to(DEV)
    flags = torch.from_numpy((g["next_state"].max(axis=1).astype(np.uint8) << 3)).to(DEV)
    rin = torch.from_numpy(g["reward_in"]).to(DEV)
    seen = ops.SeenStates(DEV, capacity_log2=4)
    got, nov = [], []
    for lo, hi in ((0, 5), (5, 1000), (1000, 1001), (1001, n)):
        o, v = ops.remember_shaping(seen, nxt[lo:hi].contiguous(), state_max[lo:hi].contiguous(), flags[lo:hi].contiguous(),
                                    rin[lo:hi].contiguous(), want_novel=True)
        got.append(o.cpu().numpy()); nov.append(v.cpu().numpy())
    assert np.array_equal(np.concatenate(got), g["reward_out"])
    assert len(seen) == int(g["final_seen"]) == int(np.concatenate(nov).sum())
    assert (1 << int(seen.highest.item())) == int(g["final_highest_tile"])
    assert seen.index == n and int(seen.overflow.item()) == 0


def test_seen_states_overflow_is_reported_and_batches_need_no_sync(g2048, oracle):
    """(1) A table that is too small for what it is given (reserve() bypassed: g2048_seen_insert called directly) sets the
    overflow flag, and the flag is reported -- by assert_ok(), by remember_shaping(check=True) for the batch itself, and at
    the next re-sizing. (2) Once the table has room under the host-side bound, further batches read nothing back."""
    from g2048 import ops, _lib as L
    n = 4096
    boards = ops.synth_boards(n, seed=31, device=DEV, p_empty=0.3, max_code=11)
    seen = ops.SeenStates(DEV, capacity_log2=6)                      # 64 slots for ~4096 distinct keys
    slots = torch.empty(n, dtype=torch.int32, device=DEV)
    L.call(seen.device, L.lib().g2048_seen_insert, boards.data_ptr(), 0, seen.table.data_ptr(), seen.capacity_log2,
           seen.count.data_ptr(), seen.overflow.data_ptr(), slots.data_ptr(), n, L.stream_ptr(seen.device))
    assert int(seen.overflow.item()) != 0 and bool((slots == -1).any())          # slot 0xffffffff = "not stored"
    with pytest.raises(RuntimeError, match="overflowed"):
        seen.assert_ok()
    mx = boards.max(dim=1).values.contiguous()                       # state = next state here: its max code, and the flags byte
    fl = (mx << 3).contiguous()                                      # a step would have written for it
    rw = torch.zeros(n, dtype=torch.float64, device=DEV)
    with pytest.raises(RuntimeError, match="overflowed"):                         # the re-sizing point looks at the flag
        ops.remember_shaping(seen, boards, mx, fl, rw)
    # (2) a fresh set: the first batches size the table (host syncs), later ones fit under the bound and enqueue only
    seen = ops.SeenStates(DEV, capacity_log2=4)
    R = oracle.Remember()
    hb = boards.cpu().numpy()
    calls = {"n": 0}
    real = torch.Tensor.item
    def counting_item(self):
        calls["n"] += 1
        return real(self)
    for rep in range(6):
        torch.Tensor.item = counting_item
        try:
            got = ops.remember_shaping(seen, boards, mx, fl, rw, check=(rep == 5))
        finally:
            torch.Tensor.item = real
        want, _ = R.batch(hb, hb, np.zeros(n))
        assert np.array_equal(got.cpu().numpy(), want)
        if rep in (3, 4):
            assert calls["n"] == 0, "a batch that fits under the host-side bound must not read anything back"
        calls["n"] = 0
    assert len(seen) == R.n_seen


def test_remember_shaping_large_ordered_batch(g2048, oracle):
    """1.2 M transitions with heavy repetition (65,536 distinct boards): first-occurrence-in-order and the running
    maximum are exact for a batch far larger than a wave / a block / a scan tile."""
    from g2048 import ops
    n, distinct = 1_200_000, 65536
    rng = np.random.default_rng(5)
    pool = oracle.synth_boards(distinct, seed=77)
    pool[:, 0] = np.maximum(pool[:, 0], 1)
    pick = rng.integers(0, distinct, n)
    nxt_h = pool[pick]
    # codes ramp up slowly so that the running maximum changes at many places of the batch
    cap = np.minimum(17, 1 + (np.arange(n) // 80000)).astype(np.uint8)
    nxt_h = np.minimum(nxt_h, cap[:, None])
    st_h = pool[rng.integers(0, distinct, n)]
    rin_h = rng.normal(size=n)
    R = oracle.Remember()
    want, wnov = R.batch(st_h, nxt_h, rin_h)
    seen = ops.SeenStates(DEV, capacity_log2=10)
    got, nov = ops.remember_shaping(seen, torch.from_numpy(nxt_h).to(DEV), torch.from_numpy(st_h.max(axis=1)).to(DEV),
                                    torch.from_numpy(nxt_h.max(axis=1) << 3).to(DEV), torch.from_numpy(rin_h).to(DEV),
                                    want_novel=True)
    assert np.array_equal(nov.cpu().numpy().astype(bool), wnov)
    assert np.array_equal(got.cpu().numpy(), want)
    assert len(seen) == R.n_seen


def test_fused_rollout_step_equals_unfused_ops(g2048):
    """g2048_rollout_step == g2048_sample_actions -> g2048_step -> g2048_obs_* -> g2048_valid_moves, bit for bit, for every
    observation dtype, f32 / f64 reward, given / recomputed mask, host step index / device counter."""
    from g2048 import ops
    torch.manual_seed(9)
    n, seed, idb = 70001, 0xFEED, 123456789
    b = ops.synth_boards(n, seed=5, device=DEV, p_empty=0.2, max_code=6)
    b[:500] = ops.synth_boards(500, seed=6, device=DEV, p_empty=0.0, max_code=3)          # many finished / nearly dead boards
    probs = torch.softmax(torch.randn(n, 4, device=DEV) * 2, 1).contiguous()
    m = ops.valid_moves(b)
    counter = torch.tensor([40], dtype=torch.int64, device=DEV)
    for odt, rdt, give_mask, use_counter in ((torch.float32, torch.float32, True, False), (torch.float16, torch.float64, False, True),
                                             (torch.bfloat16, torch.float32, True, True)):
        t = 47
        sc1 = torch.arange(n, dtype=torch.int32, device=DEV); sc2 = sc1.clone()
        a1, p1 = ops.sample_actions(probs, m, seed=seed, step_index=t, id_base=idb)
        o1, r1, f1 = ops.step(b, a1, sc1, seed, t, idb, reward_f64=rdt == torch.float64, auto_reset=True)
        obs1, m1 = ops.obs(o1, dtype=odt), ops.valid_moves(o1)
        nb_ref, _, _ = ops.step(b, a1, sc1.clone(), seed, t, idb, auto_reset=False)
        obs2 = torch.empty((n, 16), dtype=odt, device=DEV); m2 = torch.empty(n, dtype=torch.uint8, device=DEV)
        nb2 = torch.empty_like(b); sm2 = torch.empty(n, dtype=torch.uint8, device=DEV)
        o2, a2, p2, r2, f2 = ops.rollout_step(b, probs, sc2, seed, 7 if use_counter else t, idb, mask=m if give_mask else None,
                                              reward=torch.empty(n, dtype=rdt, device=DEV), obs_next=obs2, mask_next=m2,
                                              next_boards=nb2, state_maxcode=sm2, auto_reset=True,
                                              step_counter=counter if use_counter else None)
        assert bool((a1 == a2).all()) and bool((p1 == p2).all())
        assert bool((o1 == o2).all()) and bool((f1 == f2).all()) and bool((sc1 == sc2).all())
        assert np.array_equal(r1.cpu().numpy(), r2.cpu().numpy(), equal_nan=True)
        assert bool((obs1.view(torch.int16 if odt != torch.float32 else torch.int32) ==
                     obs2.view(torch.int16 if odt != torch.float32 else torch.int32)).all()) and bool((m1 == m2).all())
        assert bool((nb2 == nb_ref).all()) and bool((sm2 == b.max(dim=1).values).all())
        assert int((f1 & 1).sum()) > 0          # auto-resets happened


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


# ---------------------------------------------------------------------------------------------------------------------------
# f2: PPOMemory.sample on the device (agents/ppo_agent.py:21-50, :342-354)

def feistel_indices(batch, n, k0, k1):
    """numpy restatement of csrc/g2048_rollout.hip minibatch_index: four Feistel rounds on 2h bits, walked until below n."""
    def rnd(x, key):
        h = (x ^ np.uint32(key)).astype(np.uint32)
        h ^= h >> np.uint32(16); h = (h * np.uint32(0x7FEB352D)).astype(np.uint32)
        h ^= h >> np.uint32(15); h = (h * np.uint32(0x846CA68B)).astype(np.uint32)
        h ^= h >> np.uint32(16)
        return h
    bits = 1
    while (1 << bits) < n:
        bits += 1
    hb = (bits + 1) // 2
    mask = np.uint32((1 << hb) - 1)
    keys = [k0, k1, (k0 * 0x9E3779B1 + 1) & 0xFFFFFFFF, (k1 * 0x85EBCA77 + 2) & 0xFFFFFFFF]
    x = np.arange(batch, dtype=np.uint64)
    todo = np.ones(batch, dtype=bool)
    while todo.any():
        l = ((x[todo] >> np.uint64(hb)).astype(np.uint32)) & mask
        r = x[todo].astype(np.uint32) & mask
        for key in keys:
            l, r = r, l ^ (rnd(r, key) & mask)
        x[todo] = (l.astype(np.uint64) << np.uint64(hb)) | r.astype(np.uint64)
        todo = x >= n
    return x.astype(np.int64)


def test_minibatch_sample_is_index_select_of_the_trajectory(g2048, oracle):
    """RolloutCollector.sample(B): distinct indices, every output equal to index_select of the trajectory buffers at those
    indices, next-state observations equal to the oracle's normalize_state of the gathered (pre-auto-reset) next boards, the
    indices equal to the numpy restatement of the keyed permutation -- and no host synchronisation inside."""
    torch.manual_seed(3)
    n, T = 1024, 48
    pol = TinyTransformerPolicy().to(DEV).eval()
    for shaping in (False, True):
        rc = g2048.RolloutCollector(n, T, pol, device=DEV, seed=21, id_base=9, shaping=shaping, minibatches=True)
        with pytest.raises(RuntimeError, match="collect"):
            rc.sample(8)
        res = rc.collect()
        torch.cuda.synchronize()
        flat = lambda t: t.reshape(T * n, *t.shape[2:])        # noqa: E731
        calls = {"sync": 0, "item": 0}
        real_sync, real_item = torch.cuda.synchronize, torch.Tensor.item
        torch.cuda.synchronize = lambda *a, **k: calls.__setitem__("sync", calls["sync"] + 1) or real_sync(*a, **k)
        torch.Tensor.item = lambda self: calls.__setitem__("item", calls["item"] + 1) or real_item(self)
        try:
            batches = [rc.sample(B, want_indices=True) for B in (64, 4096, 1)]
        finally:
            torch.cuda.synchronize, torch.Tensor.item = real_sync, real_item
        assert calls == {"sync": 0, "item": 0}, "sample() must only enqueue"
        for call, mb in enumerate(batches):
            idx = mb["indices"]
            B = idx.shape[0]
            assert B == (64, 4096, 1)[call] and int(idx.unique().numel()) == B and int(idx.min()) >= 0 and int(idx.max()) < T * n
            from oracle import oracle as O
            k0, k1 = O.rng_keys(21, 9, call)                                   # DOM_MINIBATCH = 9, index = sample() call number
            assert np.array_equal(idx.cpu().numpy(), feistel_indices(B, T * n, k0, k1))
            assert torch.equal(mb["states"], flat(res["obs"]).index_select(0, idx)) and mb["states"].dtype == torch.float32
            assert torch.equal(mb["actions"], flat(res["actions"]).index_select(0, idx).to(torch.int64))
            assert torch.equal(mb["old_log_probs"], flat(res["log_prob"]).index_select(0, idx))
            rew = flat(res["shaping"]).index_select(0, idx).float() if shaping else flat(res["rewards"]).index_select(0, idx)
            assert torch.equal(mb["rewards"], rew)
            assert torch.equal(mb["dones"], flat(res["dones"]).index_select(0, idx).float())
            nb = flat(rc.next_boards).index_select(0, idx).cpu().numpy()
            assert np.array_equal(mb["next_states"].cpu().numpy(), oracle.obs_batch(nb))
        # the next states are the env's return values BEFORE auto-reset: for a finished env that is the dead (full) board, not the
        # fresh two-tile episode the trajectory's next observation row shows
        mb = batches[1]
        done = mb["dones"].cpu().numpy() > 0
        after = rc._obs[1:].reshape(T * n, 16).index_select(0, mb["indices"]).float().cpu().numpy()
        got = mb["next_states"].cpu().numpy()
        assert np.array_equal(after[~done], got[~done])
        if done.any():
            assert ((got[done] > 0).sum(axis=1) == 16).all() and ((after[done] > 0).sum(axis=1) == 2).all()
        whole = rc.sample(10 * T * n)                                           # larger than the buffer: the whole buffer, once
        assert whole["actions"].shape[0] == T * n
        a = rc.sample(32, generator=torch.Generator().manual_seed(5), want_indices=True)       # key taken from a CPU generator
        assert int(a["indices"].unique().numel()) == 32
        first = a["indices"].clone()
        b = rc.sample(32, out=a)                                                               # buffers reused: same dict, new draw
        assert b is a and int(a["indices"].unique().numel()) == 32 and not torch.equal(a["indices"], first)
        assert torch.equal(a["states"], flat(res["obs"]).index_select(0, a["indices"]))
        with pytest.raises(ValueError):
            rc.sample(33, out=a)
        if shaping:
            rc.check()
    with pytest.raises(RuntimeError, match="minibatches"):
        g2048.RolloutCollector(64, 4, pol, device=DEV).sample(4)


def test_minibatch_gather_obs_dtypes_and_f64_rewards(g2048, oracle):
    """g2048_minibatch_gather on raw arrays: float16 / bfloat16 observations widen exactly, float64 rewards round once."""
    from g2048 import ops
    n = 5000
    boards = ops.synth_boards(n, seed=77, device=DEV)
    nxt = ops.synth_boards(n, seed=78, device=DEV)
    acts = ops.synth_actions(n, seed=77, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(1)
    logp = torch.randn(n, generator=g).to(DEV)
    rew = torch.randn(n, generator=g, dtype=torch.float64).to(DEV) * 1e3
    flags = (torch.rand(n, generator=g) < 0.1).to(torch.uint8).to(DEV)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        obs = ops.obs(boards, dtype=dt)
        mb = ops.minibatch_gather(obs, acts, logp, rew, nxt, flags, 777, seed=5, sample_index=3, want_indices=True)
        idx = mb["indices"]
        assert int(idx.unique().numel()) == 777
        assert torch.equal(mb["states"], obs.index_select(0, idx).float())
        assert torch.equal(mb["rewards"], rew.index_select(0, idx).float())
        assert torch.equal(mb["dones"], flags.index_select(0, idx).float())
        assert np.array_equal(mb["next_states"].cpu().numpy(), oracle.obs_batch(nxt.index_select(0, idx).cpu().numpy()))
    one = ops.minibatch_gather(obs[:1], acts[:1], logp[:1], rew[:1], nxt[:1], flags[:1], 5, seed=1, sample_index=0, want_indices=True)
    assert one["indices"].tolist() == [0]


def test_collector_reports_a_seen_states_overflow(g2048):
    """A flag raised in the seen-states table during a run is noticed by the collector itself (every CHECK_EVERY collects, or
    check()), not only when the table next has to grow."""
    pol = TinyTransformerPolicy().to(DEV).eval()
    rc = g2048.RolloutCollector(256, 8, pol, device=DEV, seed=3, shaping=True)
    rc.CHECK_EVERY = 2
    rc.collect()
    rc.seen.overflow.fill_(2)                       # what the spin bound of g2048_seen_insert would leave behind
    with pytest.raises(RuntimeError, match="overflowed"):
        rc.collect()
    rc.seen.overflow.zero_()
    rc.collect()
    rc.seen.overflow.fill_(1)
    with pytest.raises(RuntimeError, match="overflowed"):
        rc.check()


def test_minibatch_at_config4_size(g2048):
    """BASELINE config 4's buffer (65,536 envs x 128 steps = 8,388,608 transitions): a 1 Mi-sample minibatch has distinct indices
    that cover the range evenly (a permutation prefix, not a clustered walk), equals index_select of the trajectory, and two
    consecutive sample() calls draw different permutations."""
    class Uniform(nn.Module):
        def forward(self, x):
            return torch.full((x.shape[0], 4), 0.25, device=x.device)
    n, T = 65536, 128
    rc = g2048.RolloutCollector(n, T, Uniform(), device=DEV, seed=8, minibatches=True)
    res = rc.collect()
    B = 1 << 20
    a, b = rc.sample(B, want_indices=True), rc.sample(B, want_indices=True)
    for mb in (a, b):
        idx = mb["indices"]
        assert int(idx.unique().numel()) == B and int(idx.max()) < n * T
        hist = torch.bincount(idx // (n * T // 64), minlength=64).float()
        assert float(hist.min()) > 0.9 * B / 64 and float(hist.max()) < 1.1 * B / 64
        assert torch.equal(mb["states"], res["obs"].reshape(n * T, 16).index_select(0, idx))
        assert torch.equal(mb["actions"], res["actions"].reshape(n * T).index_select(0, idx).to(torch.int64))
        assert torch.equal(mb["rewards"], res["rewards"].reshape(n * T).index_select(0, idx))
    assert not torch.equal(a["indices"], b["indices"])
    assert int(torch.isin(a["indices"][:4096], b["indices"][:4096]).sum()) < 64

"""GPU parity tests of the beam search kernel (one wavefront per game) through the C-ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.ensure_built()
    ge.import_package()
    from g2048 import ops as o
    return o


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def test_beam_golden_decisions(ops):
    """300 decisions recorded from the reference's BeamSearchAgent.get_action: (20,30) (10,15) (15,20) (3,4),
    with and without a caller mask, incl. no-valid-move, single-move and random-fallback roots."""
    g = load_golden("beam_decisions.npz")
    seed, step_index = int(g["seed"]), int(g["step_index"])
    bad = []
    for i in range(g["root"].shape[0]):
        mask = None if g["mask"][i] < 0 else torch.tensor([int(g["mask"][i])], dtype=torch.uint8, device=DEV)
        a, p = ops.beam_get_action(dev(g["root"][i:i + 1]), int(g["width"][i]), int(g["depth"][i]), mask,
                                   seed=seed, step_index=step_index, game_id_base=int(g["game_id"][i]))
        if int(a.item()) != g["action"][i] or float(p.item()) != g["prob"][i]:
            bad.append(i)
    assert not bad, bad


@pytest.mark.parametrize("width,depth", [(20, 30), (10, 15), (32, 12), (1, 6), (16, 30)])
def test_beam_batch_vs_oracle(ops, oracle, width, depth):
    """Config 3 shape: 4096 concurrent games; actions, probabilities and expansion counts vs the oracle."""
    n = 4096 if (width, depth) == (20, 30) else 1024
    hb = np.concatenate([oracle.synth_boards(n // 2, seed=51), oracle.synth_boards(n - n // 2, seed=52, p_empty=0.1, max_code=6)])
    roots = dev(hb)
    a, p, e = ops.beam_get_action(roots, width, depth, seed=99, step_index=7, game_id_base=1000, want_expanded=True)
    oa, op, oe = oracle.beam_batch(hb, width, depth, seed=99, step_index=7, game_id_base=1000)
    assert np.array_equal(a.cpu().numpy(), oa)
    assert np.array_equal(p.cpu().numpy(), op)
    assert np.array_equal(e.cpu().numpy().astype(np.uint32), oe)
    assert len(set(oa.tolist())) == 4


def test_beam_with_caller_masks_and_thresholds(ops, oracle):
    n = 2048
    hb = oracle.synth_boards(n, seed=61, p_empty=0.25, max_code=12)
    env_mask = oracle.valid_moves_batch(hb, False)
    a, p, e = ops.beam_get_action(dev(hb), 12, 18, dev(env_mask), early_threshold=256, mid_threshold=2048,
                                  seed=5, step_index=1, want_expanded=True)
    oa, op, oe = oracle.beam_batch(hb, 12, 18, mask=env_mask, early_thr=256, mid_thr=2048, seed=5, step_index=1)
    assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(p.cpu().numpy(), op)
    assert np.array_equal(e.cpu().numpy().astype(np.uint32), oe)


def test_drop_in_classes_train_loop(ops, oracle):
    """A train.py-shaped loop (reference train.py:48-107) over the look-alike classes; every transition is
    replayed through the oracle with the same draw schedule."""
    from environment.game_2048 import Game2048Env
    from agents.beam_search_agent import BeamSearchAgent
    env = Game2048Env(seed=1234)
    agent = BeamSearchAgent(beam_width=8, search_depth=10, seed=77)
    state = env.reset()
    assert state.dtype == np.int32 and state.shape == (16,)
    k0, k1 = oracle.rng_keys(1234, oracle.DOM_STEP, 0)
    score = 0
    for step in range(60):
        valid_moves = env.get_valid_moves()
        assert isinstance(valid_moves, list) and all(isinstance(v, bool) for v in valid_moves)
        assert valid_moves == [bool((oracle.env_valid_mask(state) >> a) & 1) for a in range(4)]
        action, prob = agent.get_action(state, valid_moves)
        assert isinstance(action, int) and isinstance(prob, float)
        ref = oracle.beam_get_action(state, sum(int(v) << a for a, v in enumerate(valid_moves)), 8, 10,
                                     seed=77, step_index=step, game_id=0)
        assert (action, prob) == (ref["action"], ref["prob"])
        next_state, reward, done, info = env.step(action)
        k0, k1 = oracle.rng_keys(1234, oracle.DOM_STEP, step)
        b, score, r, d, v, hi = oracle.env_step(state, score, action, oracle.rng_draw(k0, k1, 0, 0))
        assert np.array_equal(next_state, b) and reward == r and done == d
        assert isinstance(reward, np.float64) and isinstance(done, bool)
        assert info["score"] == score and info["valid_move"] == v and info["highest_tile"] == hi
        assert np.array_equal(env.board, b.reshape(4, 4)) and env.board.dtype == np.int32
        state = next_state
        if done:
            break
    env.board = np.array([[2, 2, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 4]], dtype=np.int32)
    assert env.get_valid_moves() == [True, True, True, True]
    s, r, d, info = env.step(0)
    assert s[0] == 4 and info["valid_move"]
    # an action outside 0..3 moves nothing, as in the reference (_execute_move, :97-114): invalid move, no spawn
    before, score_before = env.board.copy(), env.score
    s, r, d, info = env.step(7)
    want = oracle.env_step(before.flatten(), int(score_before), 7, 0)
    assert np.array_equal(env.board, before) and not info["valid_move"] and r == want[2] and env.score == score_before


def test_beam_arbitrary_masks_and_random_fallback(ops):
    """Golden decisions taken from the reference with adversarial caller masks (random fallback included)."""
    g = load_golden("beam_masks.npz")
    seed, si = int(g["seed"]), int(g["step_index"])
    roots = dev(g["root"][g["root_index"]])
    bad = []
    for i in range(g["mask"].shape[0]):
        m = torch.tensor([int(g["mask"][i])], dtype=torch.uint8, device=DEV)
        a, p = ops.beam_get_action(roots[i:i + 1], 5, 6, m, seed=seed, step_index=si, game_id_base=int(g["game_id"][i]))
        if int(a.item()) != g["action"][i] or float(p.item()) != g["prob"][i]:
            bad.append(i)
    assert not bad, bad


def test_beam_fixed_down_option(ops, oracle):
    """G2048_BEAM_FIXED_DOWN (true DOWN instead of the reference's rot180 quirk; NOT reference parity): checked against
    the oracle's equally modified restatement, and shown to differ from the default on some roots."""
    n = 1024
    hb = oracle.synth_boards(n, seed=81, p_empty=0.3, max_code=10)
    a, p, e = ops.beam_get_action(dev(hb), 10, 12, seed=3, step_index=2, fixed_down=True, want_expanded=True)
    oa, op, oe = oracle.beam_batch(hb, 10, 12, seed=3, step_index=2, fixed_down=True)
    assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(p.cpu().numpy(), op)
    assert np.array_equal(e.cpu().numpy().astype(np.uint32), oe)
    a0, _ = ops.beam_get_action(dev(hb), 10, 12, seed=3, step_index=2)
    assert (a0.cpu().numpy() != oa).any()


@pytest.mark.parametrize("width,depth,p_empty,max_code", [(2, 35, 0.1, 5), (5, 9, 0.6, 12), (17, 26, 0.0, 3), (32, 30, 0.25, 17),
                                                         (20, 30, 0.9, 2), (9, 3, 0.4, 9)])
def test_beam_fuzz(ops, oracle, width, depth, p_empty, max_code):
    """Differential fuzz over widths (one and two stage-B rounds), depths (incl. the depth clamps) and root
    distributions (dead / full / nearly empty boards, huge tiles), game ids above 2^32."""
    n = 2048
    hb = oracle.synth_boards(n, seed=width * 100 + depth, p_empty=p_empty, max_code=max_code)
    a, p, e = ops.beam_get_action(dev(hb), width, depth, seed=12, step_index=3, game_id_base=3 << 34, want_expanded=True)
    oa, op, oe = oracle.beam_batch(hb, width, depth, seed=12, step_index=3, game_id_base=3 << 34)
    assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(p.cpu().numpy(), op)
    assert np.array_equal(e.cpu().numpy().astype(np.uint32), oe)


@pytest.mark.parametrize("width,depth,n", [(33, 12, 512), (48, 20, 384), (64, 30, 256), (100, 8, 256), (128, 14, 192)])
def test_beam_wide_beams_vs_oracle(ops, oracle, width, depth, n):
    """The reference accepts any beam_width (agents/beam_search_agent.py:13-22): widths above 32 run stage A in rounds of
    32 parents and stage B in up to eight 64-lane passes; actions, probabilities and expansion counts vs the oracle."""
    hb = np.concatenate([oracle.synth_boards(n // 2, seed=width), oracle.synth_boards(n - n // 2, seed=width + 1, p_empty=0.05, max_code=5)])
    a, p, e = ops.beam_get_action(dev(hb), width, depth, seed=7, step_index=width, game_id_base=77, want_expanded=True)
    oa, op, oe = oracle.beam_batch(hb, width, depth, seed=7, step_index=width, game_id_base=77)
    assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(p.cpu().numpy(), op)
    assert np.array_equal(e.cpu().numpy().astype(np.uint32), oe)
    assert int(oe.max()) > 64 * 4          # more than four passes' worth of children on some level sum


@pytest.mark.parametrize("width,depth", [(1, 8), (7, 30), (16, 30), (17, 30), (20, 30), (32, 12), (50, 10), (128, 6)])
@pytest.mark.parametrize("n", [1, 777])
def test_beam_widths_dead_roots_huge_tiles_masks(ops, oracle, width, depth, n):
    """Every block shape (1 / 2 / 4 / 8 wavefronts per game) against the oracle: dead roots, single-move roots, huge tiles,
    caller masks, a single game and a ragged batch."""
    hb = np.concatenate([oracle.synth_boards(n - n // 3, seed=900 + width, p_empty=0.2, max_code=17),
                         oracle.synth_boards(n // 3, seed=901 + width, p_empty=0.0, max_code=3)])
    for mask in (None, dev(oracle.valid_moves_batch(hb, False))):
        a, p, e = ops.beam_get_action(dev(hb), width, depth, mask, seed=21, step_index=5, game_id_base=(1 << 33) + 9, want_expanded=True)
        oa, op, oe = oracle.beam_batch(hb, width, depth, mask=None if mask is None else mask.cpu().numpy(), seed=21, step_index=5,
                                       game_id_base=(1 << 33) + 9)
        assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(p.cpu().numpy(), op)
        assert np.array_equal(e.cpu().numpy().astype(np.uint32), oe)


def test_ranking_network_sorts(ops):
    """The beam kernel's bitonic network (DPP / v_permlane swaps) alone: 64 keys per wavefront sorted descending; with 16
    extra keys in the last row, the first 48 positions are the 48 largest of all 80, descending."""
    from g2048 import _lib as L
    g = torch.Generator().manual_seed(5)
    n_waves = 512
    vals = (torch.randperm(1 << 24, generator=g)[: n_waves * 80] + 1).to(torch.int64)
    keys = vals[: n_waves * 64].reshape(n_waves, 64).clone()
    extra = vals[n_waves * 64:].reshape(n_waves, 16).clone()
    keys[3, 40:] = 0                                    # "no key" lanes
    keys[4, :] = 0
    keys[5] = torch.arange(64, 0, -1)                   # already sorted / reversed inputs
    keys[6] = torch.arange(1, 65)
    extra[7, 5:] = 0                                    # fewer than 16 extra keys
    extra[8, :] = 0
    extra[9] = torch.arange(1 << 25, (1 << 25) + 16)    # every extra key beats every base key
    dk = keys.to(torch.int32).to(DEV).contiguous()
    L.call(dk.device, L.lib().g2048_sort_selftest, dk.data_ptr(), None, n_waves, 32, L.stream_ptr(dk.device))
    want = torch.sort(keys, dim=1, descending=True).values
    assert torch.equal(dk.cpu().to(torch.int64), want)
    dk = keys.to(torch.int32).to(DEV).contiguous()
    de = extra.to(torch.int32).to(DEV).contiguous()
    L.call(dk.device, L.lib().g2048_sort_selftest, dk.data_ptr(), de.data_ptr(), n_waves, 32, L.stream_ptr(dk.device))
    want = torch.sort(torch.cat([keys, extra], dim=1), dim=1, descending=True).values[:, :48]
    assert torch.equal(dk.cpu().to(torch.int64)[:, :48], want)


def test_ranking_network_sorts_64_bit_keys(ops):
    """The 64-bit variant (levels whose scores are f64): keys differing only in the high word, only in the low word, above
    2^63 and below 2^31."""
    from g2048 import _lib as L
    g = torch.Generator().manual_seed(9)
    n_waves = 256
    lo = torch.randint(0, 1 << 32, (n_waves, 80), generator=g, dtype=torch.int64)
    hi = torch.randint(0, 1 << 32, (n_waves, 80), generator=g, dtype=torch.int64)
    hi[0] = 7                                           # equal high words: the low word decides
    hi[1, :40] = 0                                      # keys below 2^32
    hi[2] = torch.randint(0, 4, (80,), generator=g)     # many equal high words
    lo[3] = 5                                           # equal low words
    hi[4] = torch.randint((1 << 32) - 4, 1 << 32, (80,), generator=g)       # top bit set: an unsigned compare is needed
    def words32(x):                                     # (n, m, 2) words in [0, 2^32) -> int32 bit patterns on the device
        x = x.reshape(x.shape[0], -1)
        return torch.where(x >= (1 << 31), x - (1 << 32), x).to(torch.int32).to(DEV).contiguous()

    for with_extra in (False, True):
        dk = words32(torch.stack([lo[:, :64], hi[:, :64]], dim=2))
        de = words32(torch.stack([lo[:, 64:], hi[:, 64:]], dim=2)) if with_extra else None
        L.call(dk.device, L.lib().g2048_sort_selftest, dk.data_ptr(), de.data_ptr() if de is not None else None, n_waves, 64,
               L.stream_ptr(dk.device))
        out = dk.cpu().to(torch.int64) & 0xffffffff
        m, keep = (80, 48) if with_extra else (64, 64)
        for w in range(n_waves):
            got = [(int(out[w, 2 * i + 1]) << 32) | int(out[w, 2 * i]) for i in range(keep)]
            want = sorted(((int(hi[w, i]) << 32) | int(lo[w, i]) for i in range(m)), reverse=True)[:keep]
            assert got == want, (w, with_extra)


@pytest.mark.parametrize("width,depth", [(20, 30), (10, 15), (16, 8), (17, 12), (32, 12), (3, 6)])
def test_rank_by_counting_switch_same_decisions(ops, oracle, width, depth):
    """G2048_BEAM_RANK_BY_COUNTING: every level ranked by the counting loop (fast levels in the plain layout, f64 levels in
    the network's layout -- the path a level takes when two scores are a few ulp apart). Same decisions and expansion
    counts as the sorting network, and as the oracle."""
    n = 1536
    roots = torch.cat([ops.synth_boards(n // 2, seed=21, id_base=0, device=DEV),
                       ops.synth_boards(n // 2, seed=22, id_base=0, p_empty=0.08, max_code=6, device=DEV)])    # crowded: ties
    a0, p0, e0 = ops.beam_get_action(roots, width, depth, seed=5, step_index=2, game_id_base=77, want_expanded=True)
    a1, p1, e1 = ops.beam_get_action(roots, width, depth, seed=5, step_index=2, game_id_base=77, want_expanded=True,
                                     rank_by_counting=True)
    assert torch.equal(a0, a1) and torch.equal(p0, p1) and torch.equal(e0, e1)
    oa, op, oe = oracle.beam_batch(roots[:128].cpu().numpy(), width, depth, seed=5, step_index=2, game_id_base=77)
    assert np.array_equal(a0[:128].cpu().numpy(), oa) and np.array_equal(e0[:128].cpu().numpy().astype(np.uint32), oe)


def test_network_near_tie_run_reaching_into_the_beam(ops, oracle):
    """Found by comparing complete evaluations: a level of this search has f64 scores that agree in all but their last
    bits, in a run of sorted neighbours that starts inside the beam and ends outside it. The later member belongs first;
    only checking neighbours inside the beam missed it (decision 3 / 2043 expansions instead of 2 / 2028)."""
    root = torch.tensor([[0, 1, 0, 2, 0, 0, 0, 2, 0, 1, 8, 4, 3, 8, 2, 1]], dtype=torch.uint8, device=DEV)
    a, p, e = ops.beam_get_action(root, 20, 30, seed=2025, step_index=262, game_id_base=472, want_expanded=True)
    oa, op, oe = oracle.beam_batch(root.cpu().numpy(), 20, 30, seed=2025, step_index=262, game_id_base=472)
    assert (int(a), int(e)) == (int(oa[0]), int(oe[0])) == (2, 2028)


@pytest.mark.parametrize("n", [4096, 5000, 9300])
def test_balanced_block_order_is_only_an_order(ops, n):
    """From 4096 games on (and with scratch) the blocks take their games in a depth-balanced order (beam_order_kernel: counting
    sort by the search depth the root's empty cells imply, dealt to the SIMDs in alternating rows, partial last row included).
    Same decisions as the caller's order, and the workspace entry point checks its scratch."""
    from g2048 import _lib as L
    roots = torch.cat([ops.synth_boards(n // 3, seed=5, id_base=0, device=DEV),
                       ops.synth_boards(n // 3, seed=6, id_base=0, p_empty=0.7, max_code=9, device=DEV),
                       ops.synth_boards(n - 2 * (n // 3), seed=7, id_base=0, p_empty=0.1, device=DEV)])
    roots = roots[torch.randperm(n, generator=torch.Generator().manual_seed(n)).to(DEV)].contiguous()
    for depth in (6, 30):                      # depth - 5 < 10 and > 10: the cost order of the three classes differs
        a, p, e = ops.beam_get_action(roots, 20, depth, seed=77, step_index=3, game_id_base=1000, want_expanded=True)
        a1, p1, e1 = ops.beam_get_action(roots, 20, depth, seed=77, step_index=3, game_id_base=1000, want_expanded=True,
                                         balanced_order=False)
        assert torch.equal(a, a1) and torch.equal(p, p1) and torch.equal(e, e1), depth
        a2, p2, e2 = ops.beam_get_action(roots, 20, depth, seed=77, step_index=3, game_id_base=1000, want_expanded=True,
                                         balanced_order="sort")            # beam_order_kernel on this call's own roots
        assert torch.equal(a, a2) and torch.equal(p, p2) and torch.equal(e, e2), depth
    mask = torch.randint(0, 16, (n,), generator=torch.Generator().manual_seed(3), dtype=torch.uint8).to(DEV)      # caller masks too
    am, pm = ops.beam_get_action(roots, 20, 6, mask, seed=78, step_index=4, game_id_base=5, balanced_order="sort")
    am1, pm1 = ops.beam_get_action(roots, 20, 6, mask, seed=78, step_index=4, game_id_base=5, balanced_order=False)
    assert torch.equal(am, am1) and torch.equal(pm, pm1)
    need = int(L.lib().g2048_beam_workspace_bytes(n))
    assert need == 4 * n and int(L.lib().g2048_beam_workspace_bytes(100)) == 0
    small = torch.empty(need - 4, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="workspace"):
        L.call(roots.device, L.lib().g2048_beam_get_action_ws, roots.data_ptr(), None, a.data_ptr(), p.data_ptr(), None, 20, 6, 512,
               1024, L.u64(1), L.u64(0), L.u64(0), n, 0, small.data_ptr(), need - 4, L.stream_ptr(roots.device))


def test_issue_priority_is_only_a_schedule(ops):
    """A launch whose blocks are all resident at once raises and lowers each wavefront's issue priority with the levels it
    has left (s_setprio; launches beyond what the chip holds do not). A schedule, not a result: one launch of 9000 games
    (no priorities) equals the same games in pieces of 4096 + 4096 + 808 (with priorities), ids and draws kept."""
    n = 9000
    roots = torch.cat([ops.synth_boards(n // 2, seed=15, id_base=0, device=DEV),
                       ops.synth_boards(n - n // 2, seed=16, id_base=0, p_empty=0.6, max_code=10, device=DEV)])
    roots = roots[torch.randperm(n, generator=torch.Generator().manual_seed(9)).to(DEV)].contiguous()
    a, p, e = ops.beam_get_action(roots, 20, 30, seed=91, step_index=7, game_id_base=300, want_expanded=True)
    for lo in range(0, n, 4096):
        hi = min(lo + 4096, n)
        a1, p1, e1 = ops.beam_get_action(roots[lo:hi].contiguous(), 20, 30, seed=91, step_index=7, game_id_base=300 + lo,
                                         want_expanded=True)
        assert torch.equal(a[lo:hi], a1) and torch.equal(p[lo:hi], p1) and torch.equal(e[lo:hi], e1), lo


def test_order_of_the_previous_call_is_only_an_order(ops, oracle):
    """Default for large batches: the blocks of a call file their games into per-class lists (g2048_beam_get_action_hist) and the
    NEXT call on the stream deals the games from them -- no order kernel. A sequence of calls with DIFFERENT roots, a change of
    the batch size in between, depths whose class costs order differently: every call equals the caller-order run; and the
    first batch against the oracle."""
    from g2048 import _lib as L
    gen = torch.Generator().manual_seed(44)
    def batch(n, k):
        r = torch.cat([ops.synth_boards(n // 3, seed=50 + k, id_base=0, device=DEV),
                       ops.synth_boards(n // 3, seed=60 + k, id_base=0, p_empty=0.7, max_code=9, device=DEV),
                       ops.synth_boards(n - 2 * (n // 3), seed=70 + k, id_base=0, p_empty=0.1, device=DEV)])
        return r[torch.randperm(n, generator=gen).to(DEV)].contiguous()
    k = 0
    for n, depth in ((4096, 30), (4096, 30), (4096, 6), (5000, 30), (5000, 30), (5000, 6), (4096, 30), (9300, 12), (9300, 12)):
        roots = batch(n, k)
        a, p, e = ops.beam_get_action(roots, 20, depth, seed=81, step_index=k, game_id_base=7 * k, want_expanded=True)
        a1, p1, e1 = ops.beam_get_action(roots, 20, depth, seed=81, step_index=k, game_id_base=7 * k, want_expanded=True,
                                         balanced_order=False)
        assert torch.equal(a, a1) and torch.equal(p, p1) and torch.equal(e, e1), (k, n, depth)
        if k == 1:
            oa, op, oe = oracle.beam_batch(roots[:600].cpu().numpy(), 20, depth, seed=81, step_index=k, game_id_base=7 * k)
            assert np.array_equal(a[:600].cpu().numpy(), oa) and np.array_equal(e[:600].cpu().numpy(), oe)
        k += 1
    # the C-ABI itself: sizes, argument checks, and a caller that breaks the contract (jumps in call_index, never clears the buffer)
    n = 4096
    need = int(L.lib().g2048_beam_history_bytes(n))
    assert need == 116992 and int(L.lib().g2048_beam_history_bytes(100)) == 0
    hist = torch.zeros(need, dtype=torch.uint8, device=DEV)
    roots = batch(n, 99)
    ref = ops.beam_get_action(roots, 20, 30, seed=5, step_index=1, want_expanded=True, balanced_order=False)
    a = torch.empty(n, dtype=torch.uint8, device=DEV); p = torch.empty(n, dtype=torch.float32, device=DEV)
    e = torch.empty(n, dtype=torch.int32, device=DEV)
    def call(idx, buf=hist, nbytes=need):
        L.call(roots.device, L.lib().g2048_beam_get_action_hist, roots.data_ptr(), None, a.data_ptr(), p.data_ptr(), e.data_ptr(), 20, 30,
               512, 1024, L.u64(5), L.u64(1), L.u64(0), n, 0, buf.data_ptr(), nbytes, idx, L.stream_ptr(roots.device))
        assert torch.equal(a, ref[0]) and torch.equal(p, ref[1]) and torch.equal(e, ref[2]), idx
    for idx in (1, 2, 3, 4, 9, 10, 11, 3, 4, 5, 1000000, 1000001, 7, 7, 7, 8):
        call(idx)
    with pytest.raises(RuntimeError, match="history"):
        call(1, nbytes=need - 4)
    with pytest.raises(RuntimeError, match="history"):
        call(0)


def test_beam_history_is_thread_safe_and_only_an_order(ops):
    """The default (balanced_order=True) block order keeps its state in a BeamHistory -- one per (device, stream) for callers
    that pass none. Two host threads issuing interleaved calls on ONE stream (sharing that history) and on TWO streams, and a
    caller-owned history shared by two threads: every result equals the caller-order run; the per-stream registry evicts."""
    import threading
    n, calls = 4096, 100
    roots = [ops.synth_boards(n, seed=600 + k, device=DEV, p_empty=0.3 + 0.05 * k, max_code=9) for k in range(3)]
    want = {}
    for k in range(3):
        for si in (0, 1):
            a, p, e = ops.beam_get_action(roots[k], 20, 12, seed=9, step_index=si, want_expanded=True, balanced_order=False)
            want[(k, si)] = (a.clone(), e.clone())
    torch.cuda.synchronize()
    errors = []

    def worker(tid, stream, history):
        try:
            with torch.cuda.stream(stream):
                for c in range(calls):
                    k, si = (c + tid) % 3, c & 1
                    a, p, e = ops.beam_get_action(roots[k], 20, 12, seed=9, step_index=si, want_expanded=True, history=history)
                    if c % 10 == 9 or c == calls - 1:
                        stream.synchronize()
                        if not (torch.equal(a, want[(k, si)][0]) and torch.equal(e, want[(k, si)][1])):
                            errors.append((tid, c))
        except Exception as exc:        # noqa: BLE001
            errors.append((tid, repr(exc)))

    cur = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    own = ops.BeamHistory(DEV)
    for streams, hist in (((cur, cur), None), ((s1, s2), None), ((s1, s1), own)):
        th = [threading.Thread(target=worker, args=(i, streams[i], hist)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        assert not errors, errors
    assert own.calls > 0 and own.n == n
    # eviction: many short-lived streams do not pin a buffer each
    for _ in range(40):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            ops.beam_get_action(roots[0], 20, 12, seed=9, step_index=0)
        st.synchronize()
    assert len(ops._BEAM_HIST.items) <= ops._BEAM_HIST.cap

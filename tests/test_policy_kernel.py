"""The fused policy kernel (csrc/policy_mlp.hip): 32-B board state -> Model(217, 14) on the matrix cores -> action.

Two kinds of check, as the MFMA fragment layouts demand:
  * EXACT: with small-integer sparse weights every intermediate value is an integer below 256, exactly
    representable in bf16 and f32, so the kernel's logits must equal an integer reference bit for bit.  This pins
    the whole wiring: weight pre-packing, permuted k order, accumulator-as-operand hand-off, observation
    construction from the packed state, output row mapping.
  * TOLERANCE (floating point, the one place on this path): random float weights against a float64 reference
    that rounds weights and hidden activations to bf16 at the same points.  Differences come only from the
    summation order inside f32 accumulation and the occasional activation that rounds the other way:
    |logit - ref| <= 2e-2 * (1 + max|ref|) and the chosen actions agree wherever the reference's margin
    between best and runner-up exceeds that tolerance.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16_round(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def _reference(obs, params, round_hidden=True):
    x = obs.astype(np.float64)
    for i, (w, b) in enumerate(params):
        y = x @ _bf16_round(w).astype(np.float64).T + b.astype(np.float64)
        if i < 4:
            y = np.maximum(y, 0.0)
            x = _bf16_round(y.astype(np.float32)).astype(np.float64) if round_hidden else y
        else:
            x = y
    return x


def _decode(logits):
    return logits[:, :4].argmax(1) * 10 + logits[:, 4:].argmax(1)


def _env(T, n, seed=3, steps=7):
    env = T.BatchedTetris(10, 40, n, seed=seed, auto_reset=True)
    rows, pieces = env.synthetic_configs(min(n, 4096))
    env.load_configs(rows, pieces)
    env.reset()
    for t in range(steps):                                   # some played boards, counters and terminal flags in the mix
        env.step(env.synthetic_actions(t), observe=False)
    return env


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    import tetris_piclim
    return tetris_piclim


@pytest.mark.parametrize("n", [1, 31, 64, 1000, 70000])
def test_exact_integer_weights(T, n):
    import torch
    rng = np.random.default_rng(n)

    def sparse(out, inn, nnz, scale=1):
        w = np.zeros((out, inn), np.float32)
        for r in range(out):
            w[r, rng.choice(inn, nnz, replace=False)] = rng.choice([-1, 1], nnz) * scale
        return w
    # cells/one-hots are 0/1; L_rem, M_rem up to 40: keep their weights zero in layer 1 so values stay small
    w1 = sparse(128, 214, 3)
    w1 = np.concatenate([w1, np.zeros((128, 3), np.float32)], 1)
    w1[::5, 216] = 1                                         # the terminal flag takes part
    w1[3, 214] = 1; w1[7, 215] = 1                           # and once each, the two counters (<= 40)
    params = [(w1, rng.integers(0, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(14, 128, 2), rng.integers(-2, 3, 14).astype(np.float32))]
    env = _env(T, n)
    obs = env.observe().cpu().numpy()
    want = _reference(obs, params, round_hidden=False)
    assert np.abs(want).max() < 256 and np.all(want == np.round(want))      # the construction really is exact
    image = torch.from_numpy(T.pack_policy(params)).to(env.device)
    logits = torch.full((n, 14), float("nan"), device=env.device)
    action = env.policy_act(image, logits=logits)
    got = logits.cpu().numpy()
    assert np.array_equal(got, want.astype(np.float32))
    assert np.array_equal(action.cpu().numpy(), _decode(want))
    assert np.array_equal(env.decode_actions(logits).cpu().numpy(), action.cpu().numpy())   # same tie rule as the decode kernel
    env.terminate()


def test_random_float_weights_within_tolerance(T):
    import torch
    n = 20000
    torch.manual_seed(1)
    model = T.PolicyMLP()
    with torch.no_grad():
        for prm in model.parameters():
            prm.normal_(0.0, 0.3)
        model.layer1.weight[:, 214:216] *= 0.05              # keep the two large-valued counters from saturating everything
    params = [(l.weight.detach().numpy(), l.bias.detach().numpy())
              for l in (model.layer1, model.layer2, model.layer3, model.layer4, model.layer5)]
    env = _env(T, n)
    obs = env.observe().cpu().numpy()
    want = _reference(obs, params)
    image = T.actor.policy_image(model, env.device)
    logits = torch.empty((n, 14), device=env.device)
    action = env.policy_act(image, logits=logits).cpu().numpy()
    got = logits.cpu().numpy().astype(np.float64)
    tol = 2e-2 * (1.0 + np.abs(want).max())
    assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), tol)
    assert np.abs(got - want).mean() <= tol / 20
    # actions agree wherever the reference's decision is clear of the tolerance
    def margin(block):
        s = np.sort(block, axis=1)
        return s[:, -1] - s[:, -2]
    clear = (margin(want[:, :4]) > 2 * tol) & (margin(want[:, 4:]) > 2 * tol)
    assert clear.mean() > 0.5
    assert np.array_equal(action[clear], _decode(want)[clear])
    assert (action == _decode(want)).mean() > 0.97
    env.terminate()


def test_fused_actor_matches_unfused_decisions_and_steps_the_env(T, oracle):
    """Actor(fused=True) drives the environment; its recorded actions replayed through the oracle give the same
    rewards, dones and boards (the environment side is bit-exact whatever the policy does)."""
    import torch
    L, M, n, seed = 10, 40, 8192, 23
    torch.manual_seed(2)
    model = T.PolicyMLP()
    with torch.no_grad():
        for prm in model.parameters():
            prm.normal_(0.0, 0.35)
        model.layer1.weight[:, 214:] = 0.0
    env = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True)
    rows, pieces = env.synthetic_configs(1024)
    env.load_configs(rows, pieces)
    env.reset()
    actor = T.Actor(env, model, use_graph=True, fused=True)
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(rows.cpu().numpy().view(np.uint16), pieces.cpu().numpy())
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    seen = set()
    for t in range(40):
        actor.step()
        a = actor.action.cpu().numpy()
        r_c, d_c = cpu.step(a)
        assert np.array_equal(actor.reward.cpu().numpy(), r_c) and np.array_equal(actor.done.cpu().numpy(), d_c), t
        seen.update(np.unique(a).tolist())
    assert len(seen) > 8
    got = {k: v.cpu().numpy() for k, v in env.packed_state().items()}
    want = cpu.get_state()
    for k, v in want.items():
        assert np.array_equal(got[k].view(np.uint16) if k == "rows" else got[k], v), k
    env.terminate()


@pytest.mark.parametrize("auto,eps", [(True, 0.0), (True, 0.15), (False, 0.3)])
def test_actor_rollout_megakernel_equals_the_step_by_step_loop(T, oracle, auto, eps):
    """tpl_actor_rollout (T iterations in one launch) == T x (tpl_policy_act, tpl_explore_actions, tpl_step):
    same actions, rewards, dones, recorded states, final boards and statistics; and the environment side equals
    the oracle fed with the recorded actions."""
    import torch
    L, M, n, seed, steps = 10, 40, 3000, 29, 33
    torch.manual_seed(4)
    model = T.PolicyMLP()
    with torch.no_grad():
        for prm in model.parameters():
            prm.normal_(0.0, 0.35)
        model.layer1.weight[:, 214:] = 0.0
    envs = []
    for _ in range(2):
        env = T.BatchedTetris(L, M, n, seed=seed, auto_reset=auto, global_offset=77, reward=(1.0, 2.0, -1.0))
        rows, pieces = env.synthetic_configs(700)
        env.load_configs(rows, pieces)
        env.reset()
        envs.append(env)
    mega, ref = envs
    image = T.actor.policy_image(model, mega.device)
    k1 = 12                                                    # two launches: the second resumes from stored state
    out1 = mega.actor_rollout(image, k1, epsilon=eps, seed=5, step0=100, record_states=True)
    out2 = mega.actor_rollout(image, steps - k1, epsilon=eps, seed=5, step0=100 + k1, record_states=True)
    out = {k: torch.cat([out1[k], out2[k]]) for k in out1}
    cpu = oracle.Env(n, L, M, 77, seed)
    cpu.set_pool(rows.cpu().numpy().view(np.uint16), pieces.cpu().numpy())
    cpu.set_options(auto_reset=auto, assign_mode=0, per_line=1.0, win=2.0, lose=-1.0)
    cpu.reset()
    explored = 0
    for t in range(steps):
        a_planes, b_planes = ref.raw_planes()
        assert torch.equal(out["states_a"][t], a_planes) and torch.equal(out["states_b"][t], b_planes), t
        if t % 8 == 0:      # a recorded state expands to the observation the environment showed at that moment
            idx = torch.arange(0, n, 7, device=ref.device)
            assert torch.equal(ref.expand_states(out["states_a"][t][idx], out["states_b"][t][idx]), ref.observe()[idx]), t
        greedy = ref.policy_act(image).clone()
        action = ref.explore_actions(greedy.clone(), eps, seed=5, step=100 + t)
        explored += int((action != greedy).sum())
        _, r, d, _ = ref.step(action, observe=False)
        assert torch.equal(out["actions"][t], action), t
        assert torch.equal(out["rewards"][t], r) and torch.equal(out["dones"][t], d), t
        r_c, d_c = cpu.step(action.cpu().numpy())
        assert np.array_equal(r.cpu().numpy(), r_c) and np.array_equal(d.cpu().numpy(), d_c), t
    if eps > 0:
        assert 0.5 * eps < explored / (steps * n) < 1.5 * eps     # the exploration rate is what was asked for
    else:
        assert explored == 0
    got = {k: v.cpu().numpy() for k, v in mega.packed_state().items()}
    want = cpu.get_state()
    for k, v in want.items():
        assert np.array_equal(got[k].view(np.uint16) if k == "rows" else got[k], v), k
    assert mega.stats() == ref.stats() == cpu.stats()
    for e in envs:
        e.terminate()


def test_actor_megakernel_at_the_benched_size(T, oracle):
    """BASELINE configs[4] at bench.py's size: 262,144 boards, L=10, M=40, auto-reset from a pool of one entry per board,
    tpl_actor_rollout (25 iterations in one launch, epsilon-greedy) against the step-by-step loop on a twin handle and the
    oracle fed with the recorded actions."""
    import torch
    L, M, n, seed, steps = 10, 40, 262144, 0, 25
    torch.manual_seed(0)
    model = T.PolicyMLP()
    envs = []
    for _ in range(2):
        env = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, assign="hash")
        rows, pieces = env.synthetic_configs(n)
        env.load_configs(rows, pieces)
        env.reset()
        envs.append(env)
    mega, ref = envs
    image = T.actor.policy_image(model, mega.device)
    out = mega.actor_rollout(image, steps, epsilon=0.1, seed=7, step0=0)
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(rows.cpu().numpy().view(np.uint16), pieces.cpu().numpy())
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    for t in range(steps):
        action = ref.explore_actions(ref.policy_act(image).clone(), 0.1, seed=7, step=t)
        _, r, d, _ = ref.step(action, observe=False)
        assert torch.equal(out["actions"][t], action), t
        assert torch.equal(out["rewards"][t], r) and torch.equal(out["dones"][t], d), t
        r_c, d_c = cpu.step(action.cpu().numpy())
        assert np.array_equal(r.cpu().numpy(), r_c) and np.array_equal(d.cpu().numpy(), d_c), t
    got = {k: v.cpu().numpy() for k, v in mega.packed_state().items()}
    want = cpu.get_state()
    for k, v in want.items():
        assert np.array_equal(got[k].view(np.uint16) if k == "rows" else got[k], v), k
    assert mega.stats() == ref.stats() == cpu.stats() and mega.stats()["episodes"] > n
    for e in envs:
        e.terminate()


# ------------------------------------------------------------------------------------------------- float32 operands
@pytest.mark.parametrize("n", [1, 31, 64, 1000, 70000])
def test_f32_kernel_exact_integer_weights(T, n):
    """The float32 kernel (csrc/policy_f32.hip) on the same exact construction: every intermediate value is a small
    integer, so the logits must equal the integer reference bit for bit -- this pins its own wiring (k order of the
    accumulator hand-off, weight packing, chunk streaming through LDS, features from the packed state)."""
    import torch
    rng = np.random.default_rng(100 + n)

    def sparse(out, inn, nnz):
        w = np.zeros((out, inn), np.float32)
        for r in range(out):
            w[r, rng.choice(inn, nnz, replace=False)] = rng.choice([-1, 1], nnz)
        return w
    w1 = np.concatenate([sparse(128, 214, 3), np.zeros((128, 3), np.float32)], 1)
    w1[::5, 216] = 1
    w1[3, 214] = 1; w1[7, 215] = -1
    params = [(w1, rng.integers(0, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(14, 128, 2), rng.integers(-2, 3, 14).astype(np.float32))]
    env = _env(T, n)
    obs = env.observe().cpu().numpy()
    want = _reference(obs, params, round_hidden=False)
    assert np.all(want == np.round(want))
    image = torch.from_numpy(T.pack_policy(params, f32=True)).to(env.device)
    logits = torch.full((n, 14), float("nan"), device=env.device)
    action = env.policy_act(image, logits=logits)
    assert np.array_equal(logits.cpu().numpy(), want.astype(np.float32))
    assert np.array_equal(action.cpu().numpy(), _decode(want))
    env.terminate()


def test_f32_kernel_against_a_float32_torch_module(T):
    """The reference's policy is a float32 nn.Linear stack (model/model.py:9-20).  The float32 kernel against that
    very module in torch float32 AND a float64 evaluation of the same weights: what separates them is the order of
    float32 summation only.  Tolerance: |logit - ref64| <= 2e-5 * (1 + max|ref64|); the torch float32 result must sit
    inside the same band (it is no closer to float64 than the kernel is), and the chosen actions agree wherever the
    float64 margin between best and runner-up exceeds twice the tolerance.  The bf16 kernel on the same weights misses
    this band by three orders of magnitude."""
    import torch
    n = 30000
    torch.manual_seed(3)
    model = T.PolicyMLP()                                        # torch's default initialisation, as the reference would
    with torch.no_grad():
        model.layer5.weight.mul_(8.0)                            # spread the logits out
    env = _env(T, n)
    obs = env.observe()
    with torch.no_grad():
        ref32 = model.to(env.device)(obs).cpu().numpy().astype(np.float64)
        ref64 = model.double()(obs.double()).cpu().numpy()
    model = model.float().cpu()
    logits = torch.empty((n, 14), device=env.device)
    action = env.policy_act(T.actor.policy_image(model, env.device, f32=True), logits=logits).cpu().numpy()
    got = logits.cpu().numpy().astype(np.float64)
    tol = 2e-5 * (1.0 + np.abs(ref64).max())
    assert np.abs(got - ref64).max() <= tol, (np.abs(got - ref64).max(), tol)
    assert np.abs(ref32 - ref64).max() <= tol
    def margin(block):
        srt = np.sort(block, axis=1)
        return srt[:, -1] - srt[:, -2]
    clear = (margin(ref64[:, :4]) > 2 * tol) & (margin(ref64[:, 4:]) > 2 * tol)
    assert clear.mean() > 0.99 and np.array_equal(action[clear], _decode(ref64)[clear])
    lg16 = torch.empty((n, 14), device=env.device)
    env.policy_act(T.actor.policy_image(model, env.device), logits=lg16)
    assert np.abs(lg16.cpu().numpy() - ref64).max() > 100 * tol            # bf16 operands are not "the same arithmetic"
    env.terminate()


def test_f32_actor_drives_the_environment(T, oracle):
    """Actor(fused=True, dtype=float32): float32 policy kernel -> step kernel; the recorded actions replayed through the
    oracle give the same rewards, dones and boards, and equal what the float32 torch module decides where its margin is
    clear."""
    import torch
    L, M, n, seed = 10, 40, 8192, 23
    torch.manual_seed(2)
    model = T.PolicyMLP()
    env = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True)
    rows, pieces = env.synthetic_configs(1024)
    env.load_configs(rows, pieces)
    env.reset()
    actor = T.Actor(env, model, dtype=torch.float32, use_graph=False, fused=True)
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(rows.cpu().numpy().view(np.uint16), pieces.cpu().numpy())
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    agree = []
    for t in range(30):
        with torch.no_grad():
            lg = actor.model(env.observe()).cpu().numpy()
        actor.step()
        a = actor.action.cpu().numpy()
        agree.append((a == _decode(lg)).mean())
        r_c, d_c = cpu.step(a)
        assert np.array_equal(actor.reward.cpu().numpy(), r_c) and np.array_equal(actor.done.cpu().numpy(), d_c), t
    assert min(agree) > 0.995
    got = {k: v.cpu().numpy() for k, v in env.packed_state().items()}
    for k, v in cpu.get_state().items():
        assert np.array_equal(got[k].view(np.uint16) if k == "rows" else got[k], v), k
    env.terminate()


@pytest.mark.parametrize("kind", [True, "split"])
@pytest.mark.parametrize("auto,eps,n,steps", [(True, 0.0, 3000, 21), (True, 0.15, 70001, 7), (False, 0.3, 3000, 21)])
def test_f32_actor_rollout_megakernel_equals_the_step_by_step_loop(T, oracle, auto, eps, n, steps, kind):
    """tpl_actor_rollout_f32 (T iterations of float32 policy -> epsilon-greedy -> step in ONE launch, the reference's
    nn.Linear width as a multi-step loop) == T x (tpl_policy_act_f32, tpl_explore_actions, tpl_step) on a twin handle:
    same actions, rewards, dones, recorded states, final boards and statistics -- the float32 products are the same
    instructions in the same order, so the logits, and with them the decisions, are identical -- and the environment side
    equals the oracle fed with the recorded actions.  70,001 boards: more than one pass per wave, a ragged last tile.
    kind = "split": the same for tpl_actor_rollout_split against tpl_policy_act_split (three bf16 pieces per number)."""
    import torch
    L, M, seed = 10, 40, 29
    torch.manual_seed(4)
    model = T.PolicyMLP()
    with torch.no_grad():
        for prm in model.parameters():
            prm.normal_(0.0, 0.35)
    envs = []
    for _ in range(2):
        env = T.BatchedTetris(L, M, n, seed=seed, auto_reset=auto, global_offset=77, reward=(1.0, 2.0, -1.0))
        rows, pieces = env.synthetic_configs(700)
        env.load_configs(rows, pieces)
        env.reset()
        envs.append(env)
    mega, ref = envs
    image = T.actor.policy_image(model, mega.device, f32=kind)
    k1 = steps // 3                                            # two launches: the second resumes from stored state
    out1 = mega.actor_rollout(image, k1, epsilon=eps, seed=5, step0=100, record_states=True)
    out2 = mega.actor_rollout(image, steps - k1, epsilon=eps, seed=5, step0=100 + k1, record_states=True)
    out = {k: torch.cat([out1[k], out2[k]]) for k in out1}
    cpu = oracle.Env(n, L, M, 77, seed)
    cpu.set_pool(rows.cpu().numpy().view(np.uint16), pieces.cpu().numpy())
    cpu.set_options(auto_reset=auto, assign_mode=0, per_line=1.0, win=2.0, lose=-1.0)
    cpu.reset()
    explored = 0
    for t in range(steps):
        a_planes, b_planes = ref.raw_planes()
        assert torch.equal(out["states_a"][t], a_planes) and torch.equal(out["states_b"][t], b_planes), t
        greedy = ref.policy_act(image).clone()
        action = ref.explore_actions(greedy.clone(), eps, seed=5, step=100 + t)
        explored += int((action != greedy).sum())
        _, r, d, _ = ref.step(action, observe=False)
        assert torch.equal(out["actions"][t], action), t
        assert torch.equal(out["rewards"][t], r) and torch.equal(out["dones"][t], d), t
        r_c, d_c = cpu.step(action.cpu().numpy())
        assert np.array_equal(r.cpu().numpy(), r_c) and np.array_equal(d.cpu().numpy(), d_c), t
    if eps > 0:
        assert 0.5 * eps < explored / (steps * n) < 1.5 * eps
    else:
        assert explored == 0
    got = {k: v.cpu().numpy() for k, v in mega.packed_state().items()}
    want = cpu.get_state()
    for k, v in want.items():
        assert np.array_equal(got[k].view(np.uint16) if k == "rows" else got[k], v), k
    assert mega.stats() == ref.stats() == cpu.stats()
    for e in envs:
        e.terminate()


# ------------------------------------------------------------------------------------------------- three bf16 pieces
@pytest.mark.parametrize("n", [1, 31, 64, 1000, 70000])
def test_split_kernel_exact_integer_weights(T, n):
    """The split kernel (csrc/policy_split.hip: float32 accuracy from three bf16 pieces per number, six bf16 MFMAs per
    product) on the exact construction: small integers are their own high piece, the low pieces are zero, every product and
    sum is exact -- the logits must equal the integer reference bit for bit.  Pins its wiring: three weight planes streamed
    through LDS in ten chunks, k-steps outermost, the float32 hand-off split as it is used."""
    import torch
    rng = np.random.default_rng(300 + n)

    def sparse(out, inn, nnz):
        w = np.zeros((out, inn), np.float32)
        for r in range(out):
            w[r, rng.choice(inn, nnz, replace=False)] = rng.choice([-1, 1], nnz)
        return w
    w1 = np.concatenate([sparse(128, 214, 3), np.zeros((128, 3), np.float32)], 1)
    w1[::5, 216] = 1
    w1[3, 214] = 1; w1[7, 215] = -1
    params = [(w1, rng.integers(0, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(128, 128, 2), rng.integers(-1, 2, 128).astype(np.float32)),
              (sparse(14, 128, 2), rng.integers(-2, 3, 14).astype(np.float32))]
    env = _env(T, n)
    obs = env.observe().cpu().numpy()
    want = _reference(obs, params, round_hidden=False)
    assert np.all(want == np.round(want))
    image = torch.from_numpy(T.pack_policy(params, f32="split")).to(env.device)
    logits = torch.full((n, 14), float("nan"), device=env.device)
    action = env.policy_act(image, logits=logits)
    assert np.array_equal(logits.cpu().numpy(), want.astype(np.float32))
    assert np.array_equal(action.cpu().numpy(), _decode(want))
    env.terminate()


def test_split_kernel_meets_the_float32_tolerance(T):
    """The test the float32 kernel has to pass (test_f32_kernel_against_a_float32_torch_module), on the split kernel: within
    2e-5 * (1 + max|ref64|) of a float64 evaluation of the float32 weights, actions equal wherever the float64 margin is
    clear; and it is as close to float64 as the float32 MFMA kernel is (same order of magnitude of the worst error), where
    the plain bf16 kernel misses the band by more than a factor of a hundred."""
    import torch
    n = 30000
    torch.manual_seed(3)
    model = T.PolicyMLP()
    with torch.no_grad():
        model.layer5.weight.mul_(8.0)                            # spread the logits out
    env = _env(T, n)
    obs = env.observe()
    with torch.no_grad():
        ref64 = model.double().to(env.device)(obs.double()).cpu().numpy()
    model = model.float().cpu()
    logits = torch.empty((n, 14), device=env.device)
    action = env.policy_act(T.actor.policy_image(model, env.device, f32="split"), logits=logits).cpu().numpy()
    got = logits.cpu().numpy().astype(np.float64)
    tol = 2e-5 * (1.0 + np.abs(ref64).max())
    err_split = np.abs(got - ref64).max()
    assert err_split <= tol, (err_split, tol)
    lg32 = torch.empty((n, 14), device=env.device)
    env.policy_act(T.actor.policy_image(model, env.device, f32=True), logits=lg32)
    err_f32 = np.abs(lg32.cpu().numpy() - ref64).max()
    assert err_split <= 4 * err_f32 + 1e-7, (err_split, err_f32)

    def margin(block):
        srt = np.sort(block, axis=1)
        return srt[:, -1] - srt[:, -2]
    clear = (margin(ref64[:, :4]) > 2 * tol) & (margin(ref64[:, 4:]) > 2 * tol)
    assert clear.mean() > 0.99 and np.array_equal(action[clear], _decode(ref64)[clear])
    env.terminate()

"""hipGraph replay of the forward (graph.GraphedForward): bitwise equal to the eager forward, static shapes enforced, the
reference's IndexError behaviour kept in front of the replay."""
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import rosettafold_pytorch_amd as R  # noqa: E402

DEV = "cuda"
CFG = dict(d_msa=96, d_pair=64, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
           n_encoder_layers=1, max_len=80, n_neighbors=[16, 16], p_dropout=0.0)


def inputs(seed, B=2, N=8, L=64):
    g = torch.Generator().manual_seed(seed)
    msa = torch.randint(0, 21, (B, N, L), generator=g)
    return msa.to(DEV), msa[:, 0].clone().to(DEV), torch.arange(L).repeat(B, 1).to(DEV)


def flat(out):
    return [out[0][k] for k in sorted(out[0])] + [out[1], out[2]]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_graph_replay_equals_eager(dtype):
    R.set_compute_dtype(dtype)
    try:
        torch.manual_seed(11)
        model = R.RoseTTAFold(**CFG).to(DEV)
        a, b = inputs(0), inputs(1)
        ref_a = [t.clone() for t in flat(model(*a))]
        ref_b = [t.clone() for t in flat(model(*b))]
        g = R.GraphedForward(model, *a)
        for _ in range(2):
            assert all(torch.equal(x, y) for x, y in zip(flat(g(*a)), ref_a))
            assert all(torch.equal(x, y) for x, y in zip(flat(g(*b)), ref_b))   # new data through the static inputs
    finally:
        R.set_compute_dtype(torch.bfloat16)


def test_graph_keeps_the_boundary_checks():
    torch.manual_seed(11)
    model = R.RoseTTAFold(**CFG).to(DEV)
    a = inputs(0)
    g = R.GraphedForward(model, *a)
    good = [t.clone() for t in flat(g(*a))]
    bad = a[0].clone()
    bad[0, 0, 3] = 21
    with pytest.raises(IndexError):
        g(bad, a[1], a[2])
    with pytest.raises(ValueError, match="captured for"):
        g(*inputs(0, L=32))
    R.set_compute_dtype(torch.float16)       # the graph holds the other library's kernels: refused until recaptured
    try:
        with pytest.raises(R._lib.RfmiError, match="recapture"):
            g(*a)
    finally:
        R.set_compute_dtype(torch.bfloat16)
    assert all(torch.equal(x, y) for x, y in zip(flat(g(*a)), good))   # a refused call leaves the graph usable
    # unordered residue indices: the structure track needs the general edge capacity -> re-captured, equals eager
    aa = a[2].clone()
    aa[:, 10] = 5
    ref = [t.clone() for t in flat(model(a[0], a[1], aa))]
    assert all(torch.equal(x, y) for x, y in zip(flat(g(a[0], a[1], aa)), ref))


@pytest.mark.parametrize("how", ["load_state_dict", "copy_", "invalidate", "to"])
def test_graph_follows_weight_changes(how):
    """The recorded kernels read the prepared 16-bit weight copies by raw pointer; any route that drops or outdates those
    copies must be followed by a new capture, not by a replay of freed / stale memory (round-3 advisor finding)."""
    torch.manual_seed(11)
    model = R.RoseTTAFold(**CFG).to(DEV)
    a = inputs(0)
    g = R.GraphedForward(model, *a)
    before = [t.clone() for t in flat(g(*a))]
    torch.manual_seed(12)
    other = R.RoseTTAFold(**CFG).to(DEV)
    if how == "load_state_dict":
        model.load_state_dict(other.state_dict())
    elif how == "copy_":
        with torch.no_grad():
            for p, q in zip(model.parameters(), other.parameters()):
                p.copy_(q)
    elif how == "invalidate":   # same weights, copies dropped: the replay must not touch the freed copies
        R.invalidate_weight_caches(model)
        junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(64)]   # re-use the freed blocks
        del junk
    else:
        model.to(torch.float64).to(torch.float32)
    got = [t.clone() for t in flat(g(*a))]
    want = flat(model(*a))
    assert all(torch.equal(x, y) for x, y in zip(got, want))
    if how in ("load_state_dict", "copy_"):
        assert not all(torch.equal(x, y) for x, y in zip(got, before))
    else:
        assert all(torch.isfinite(x).all() for x in got)

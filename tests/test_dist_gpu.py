"""N>1 path on the GPU box: world_size 2, both ranks on cuda:0, gloo rendezvous (RCCL needs one GPU per rank).
`shard.forward_sharded` slices a global batch, runs the local forwards through librfmi.so and gathers on rank 0; the
gathered tensors must equal the single-process forward of the same batch BIT FOR BIT (samples are independent and the
kernels are deterministic)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

CFG = dict(d_input=21, d_msa=96, d_pair=72, d_node=8, d_edge=8, d_state=8, n_two_track_blocks=1, n_three_track_blocks=2,
           n_encoder_layers=1, max_len=64, n_neighbors=[16, 16], p_dropout=0.0)
B, N, L = 3, 8, 32  # 3 samples over 2 ranks: uneven shards (2 + 1)


def _inputs():
    g = torch.Generator().manual_seed(0)
    msa = torch.randint(0, 21, (B, N, L), generator=g)
    return msa, msa[:, 0].clone(), torch.arange(L).unsqueeze(0).repeat(B, 1)


def _worker(rank, world, port, path, ckpt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rosettafold_pytorch_amd as R
    from rosettafold_pytorch_amd import shard
    model = R.RoseTTAFold(**CFG)
    R.load_checkpoint(model, ckpt)  # every rank loads the same checkpoint (CPU init is not thread-count invariant)
    model = model.to("cuda:0")
    res = shard.forward_sharded(model, *_inputs(), dst=0)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"logits": {k: v.cpu() for k, v in res[0].items()}, "xyz": res[1].cpu(), "plddt": res[2].cpu(),
                    "weights": {k: v.double().sum().item() for k, v in model.state_dict().items()},
                    "dtype": str(R.RT.dtype)}, path)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_forward_sharded_world2_matches_single_process(tmp_path):
    path, ckpt = str(tmp_path / "gathered.pt"), str(tmp_path / "weights.pt")
    import rosettafold_pytorch_amd as R
    torch.manual_seed(1234)
    model = R.RoseTTAFold(**CFG)
    R.save_checkpoint(model, ckpt)
    model = model.to("cuda:0")
    ctx = mp.get_context("spawn")
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, ckpt)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = torch.load(path)
    msa, seq, aa = _inputs()
    assert got["dtype"] == str(R.RT.dtype)
    wdiff = [k for k, v in model.state_dict().items() if v.double().sum().item() != got["weights"][k]]
    assert not wdiff, ("the ranks and this process initialised different weights", wdiff[:5])
    # single process, the same per-sample batches the ranks saw (a batch of 2 and a batch of 1)
    parts = [model(msa[lo:hi].cuda(), seq[lo:hi].cuda(), aa[lo:hi].cuda()) for lo, hi in ((0, 2), (2, 3))]
    for k in got["logits"]:
        assert torch.equal(got["logits"][k], torch.cat([p[0][k] for p in parts]).cpu()), k
    assert torch.equal(got["xyz"], torch.cat([p[1] for p in parts]).cpu())
    assert torch.equal(got["plddt"], torch.cat([p[2] for p in parts]).cpu())
    print("\n[dist gpu] world_size 2 on cuda:0: gathered results == single-process forwards, bit for bit")

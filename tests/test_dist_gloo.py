"""CPU: the N>1 path (shard -> per-rank results -> all_gather -> dataset order) with world_size 2 on gloo."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probpose_code_amd.dist import ResultGather, interleave, shard_indices

    K = 17
    idx = shard_indices(n, rank, world)
    per = (n + world - 1) // world
    # fake per-rank engine output whose values encode the dataset index; the short rank hands over FEWER rows
    # (round_up=False shards are uneven) - the padding is ResultGather's business, not the caller's
    ids = torch.tensor(idx, dtype=torch.float64)
    m = len(idx)
    out = dict(
        keypoints=ids[:, None, None].expand(m, K, 2).clone(),
        scores=ids[:, None].expand(m, K).float().clone(),
        scalars=ids[None, :, None].expand(4, m, K).float().clone(),
    )
    g = ResultGather(per, K, "cpu", world)
    g(dict(keypoints=torch.full((per, K, 2), 99.0, dtype=torch.float64), scores=torch.full((per, K), 99.0),
           scalars=torch.full((4, per, K), 99.0)))  # an earlier, full step: its rows must not survive below
    host = g(out)
    g.wait()
    ok = g.counts == [len(shard_indices(n, r, world)) for r in range(world)]
    ordered = interleave(host, n)
    ok = ok and torch.equal(ordered[:, 0, 0], torch.arange(n, dtype=torch.float64)) and ordered.shape == (n, K, 7)
    ok = ok and torch.equal(ordered[:, 3, 5], torch.arange(n, dtype=torch.float64))
    ok = ok and torch.equal(g.ordered(), ordered)
    if m < per:
        ok = ok and bool((g.gathered[rank, m:] == 0).all())
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gather_restores_dataset_order(lib_built):
    world, n = 2, 7  # uneven: round_up=False leaves rank 1 one sample short
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), n, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_indices_match_default_sampler():
    from probpose_code_amd.dist import shard_indices

    assert shard_indices(10, 0, 4) == [0, 4, 8] and shard_indices(10, 3, 4) == [3, 7]
    assert sorted(sum((shard_indices(513, r, 8) for r in range(8)), [])) == list(range(513))


@pytest.mark.gpu
def test_pack_records_kernel_matches_the_torch_form():
    """pp_pack_records (one launch) against the cat / permute / cast form the CPU path uses."""
    from probpose_code_amd.dist import ResultGather, pack_records

    B, K = 5, 17
    gen = torch.Generator(device="cuda").manual_seed(3)
    out = dict(keypoints=torch.rand(B, K, 2, dtype=torch.float64, device="cuda", generator=gen),
               scores=torch.rand(B, K, device="cuda", generator=gen), scalars=torch.rand(4, B, K, device="cuda", generator=gen))
    ref = pack_records({k: v.cpu() for k, v in out.items()})
    assert torch.equal(pack_records(out).cpu(), ref)
    g = ResultGather(B, K, "cuda", 1)
    host = g(out)
    torch.cuda.synchronize()
    assert torch.equal(host[0], ref)


@pytest.mark.gpu
def test_rccl_all_gather_path_on_one_rank():
    """The box the GPU tests run on has ONE MI355X, so the N > 1 exchange cannot run there; what can is the RCCL call itself:
    a 1-rank `nccl` (= RCCL on ROCm) process group and ResultGather's collective branch forced on - the library loads, the
    communicator initialises, `all_gather_into_tensor` of the (B + 1, K, 7) float64 record runs on the device, and the
    padding / count row arrives."""
    from probpose_code_amd.dist import ResultGather, pack_records

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B, K = 6, 17
        gen = torch.Generator(device="cuda").manual_seed(5)
        out = dict(keypoints=torch.rand(4, K, 2, dtype=torch.float64, device="cuda", generator=gen),
                   scores=torch.rand(4, K, device="cuda", generator=gen), scalars=torch.rand(4, 4, K, device="cuda", generator=gen))
        g = ResultGather(B, K, "cuda", 1, force_collective=True)
        host = g(out)
        g.wait()
        assert g.counts == [4]
        assert torch.equal(host[0, :4], pack_records({k: v.cpu() for k, v in out.items()}))
        assert bool((host[0, 4:] == 0).all())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_pipelined_steps_gather_over_rccl_on_one_rank():
    """bench.py --gpus N keeps two steps in flight on every rank: the all_gather of a record is then issued from alternating
    HIP streams. With one GPU on the box the communicator has one rank, but the calls are the real ones: 1-rank `nccl`
    (= RCCL) group, StepPipeline depth 2 with the collective forced on, six different batches back to back - every record
    must be the one the serial path produces (torch's process group orders the collectives on its own stream)."""
    from probpose_code_amd import synthetic as S
    from probpose_code_amd.dist import pack_records
    from probpose_code_amd.engine import ProbPoseEngine
    from probpose_code_amd.pipeline import StepPipeline

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B, flip = 16, S.COCO_FLIP_INDICES
        eng = ProbPoseEngine(S.synthetic_state_dict("small", seed=0, logit_scale=2.0), 12, precision="bf16", device="cuda:0")
        batches = [S.synthetic_crops(B, seed=300 + i).cuda() for i in range(6)]
        want = [pack_records(eng.forward(x, True, flip)).cpu().clone() for x in batches]
        pipe = StepPipeline(eng, B, flip, depth=2, world=1, force_collective=True)
        assert all(g.collective for g in pipe.gathers)
        tickets = []
        for i, x in enumerate(batches):
            tickets.append(pipe.submit(x))
            if i >= 1:
                assert torch.equal(pipe.result(tickets[i - 1])[0], want[i - 1])
        assert torch.equal(pipe.result(tickets[-1])[0], want[-1])
        assert pipe.gather_of(tickets[-1]).counts == [B]
    finally:
        dist.destroy_process_group()

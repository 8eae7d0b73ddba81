"""StepPipeline (probpose_code_amd/pipeline.py): consecutive batches in flight on separate streams / workspaces / graphs.
The pipelined path is what bench.py times, so it is checked like the serial one: every batch's record must be the record
the serial graph replay (and the eager launch sequence) produces for that batch, bit for bit, also when the batches differ
from step to step (a stale capture, a slot reading another slot's workspace or a record overwritten before it was read
would all show)."""
import numpy as np
import pytest
import torch

from probpose_code_amd import synthetic as S
from probpose_code_amd.dist import pack_records
from probpose_code_amd.pipeline import StepPipeline


class _CpuStub:
    """CPU stand-in (the ticket / slot bookkeeping needs no GPU): keypoints encode the batch's first byte and the slot."""
    K, device = 17, torch.device("cpu")

    def forward(self, crops, flip_test=True, flip_indices=None, slot=0):
        B = crops.shape[0]
        kp = torch.full((B, self.K, 2), float(crops.flatten()[0]), dtype=torch.float64)
        kp[:, :, 1] = slot
        return dict(keypoints=kp, scores=torch.zeros(B, self.K), scalars=torch.zeros(4, B, self.K))

    forward_graph = forward


def test_tickets_slots_and_stale_results_cpu():
    pipe = StepPipeline(_CpuStub(), 4, S.COCO_FLIP_INDICES, depth=2)
    batches = [torch.full((4, 3, 8, 8), v, dtype=torch.uint8) for v in (3, 5, 7)]
    t0, t1 = pipe.submit(batches[0]), pipe.submit(batches[1])
    assert (t0, t1) == (0, 1)
    r0, r1 = pipe.result(t0).clone(), pipe.result(t1).clone()
    assert r0.shape == (1, 4, 17, 7)
    assert float(r0[0, 0, 0, 0]) == 3 and float(r0[0, 0, 0, 1]) == 0
    assert float(r1[0, 0, 0, 0]) == 5 and float(r1[0, 0, 0, 1]) == 1
    t2 = pipe.submit(batches[2])  # reuses slot 0
    assert float(pipe.result(t2)[0, 0, 0, 0]) == 7
    with pytest.raises(RuntimeError, match="no longer"):
        pipe.result(t0)
    pipe.drain()
    with pytest.raises(ValueError):
        StepPipeline(_CpuStub(), 4, S.COCO_FLIP_INDICES, depth=0)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,depth", [("bf16", 2), ("bf16", 3), ("f16x3", 2)])
def test_pipelined_batches_equal_serial_replays(precision, depth):
    from probpose_code_amd.engine import ProbPoseEngine

    B, flip = 64, S.COCO_FLIP_INDICES
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision=precision, device="cuda:0")
    batches = [S.synthetic_crops(B, seed=200 + i).cuda() for i in range(5)]
    # serial references: eager launches and the slot-0 graph, one batch at a time
    want = []
    for x in batches:
        out = eng.forward(x, True, flip)
        rec = pack_records(out).cpu().numpy().copy()
        out_g = eng.forward_graph(x, True, flip)
        rec_g = pack_records(out_g).cpu().numpy().copy()
        assert np.array_equal(rec, rec_g), "graph replay differs from the eager launch sequence"
        want.append(rec)
    assert not np.array_equal(want[0], want[1]), "the test batches must differ"
    pipe = StepPipeline(eng, B, flip, depth=depth)
    # all five in flight as fast as the host can submit, collected late (within the depth window)
    got, tickets = {}, []
    for i, x in enumerate(batches):
        tickets.append(pipe.submit(x))
        if i >= depth - 1:
            t = tickets[i - (depth - 1)]
            got[t] = pipe.result(t)[0].numpy().copy()
    for t in tickets[len(batches) - (depth - 1):]:
        got[t] = pipe.result(t)[0].numpy().copy()
    for i, t in enumerate(tickets):
        assert np.array_equal(got[t], want[i]), f"batch {i}: pipelined record differs from the serial replay"
    # and again over the same slots with the batches in another order (stale captures / leftovers from the first round)
    order = [3, 0, 4, 1, 2]
    for i in order:
        t = pipe.submit(batches[i])
        assert np.array_equal(pipe.result(t)[0].numpy(), want[i])
    # device-side outputs of the last batch are the slot's own buffers
    torch.cuda.synchronize()
    dev_out = pipe.device_outputs(t)
    assert np.array_equal(pack_records(dev_out).cpu().numpy(), want[order[-1]])


@pytest.mark.gpu
def test_depth_one_is_the_plain_replay():
    from probpose_code_amd.engine import ProbPoseEngine

    B, flip = 8, S.COCO_FLIP_INDICES
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision="bf16", device="cuda:0")
    x = S.synthetic_crops(B, seed=7).cuda()
    want = pack_records(eng.forward(x, True, flip)).cpu().numpy().copy()
    pipe = StepPipeline(eng, B, flip, depth=1)
    assert pipe.streams == [None]
    assert np.array_equal(pipe.result(pipe.submit(x))[0].numpy(), want)


@pytest.mark.gpu
def test_host_batches_take_the_copy_stream_and_give_the_same_records():
    """Batches handed over in pinned HOST memory: the H2D copy is enqueued on the copy stream before the call blocks on the
    slot (so it hides under the kernels in flight); records must equal those of device-resident batches, also when the
    slots and their staging buffers are reused."""
    from probpose_code_amd.engine import ProbPoseEngine

    B, flip = 32, S.COCO_FLIP_INDICES
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision="bf16", device="cuda:0")
    host = [S.synthetic_crops(B, seed=400 + i).pin_memory() for i in range(6)]
    want = [pack_records(eng.forward(x.cuda(), True, flip)).cpu().numpy().copy() for x in host]
    pipe = StepPipeline(eng, B, flip, depth=2)
    tickets = []
    for i, x in enumerate(host):
        tickets.append(pipe.submit(x))
        if i >= 1:
            assert np.array_equal(pipe.result(tickets[i - 1])[0].numpy(), want[i - 1]), f"host batch {i - 1}"
    assert np.array_equal(pipe.result(tickets[-1])[0].numpy(), want[-1])
    # device and host batches interleaved through the same slots
    t_dev = pipe.submit(host[2].cuda())
    t_host = pipe.submit(host[4])
    assert np.array_equal(pipe.result(t_dev)[0].numpy(), want[2]) and np.array_equal(pipe.result(t_host)[0].numpy(), want[4])
    pipe1 = StepPipeline(eng, B, flip, depth=1)
    assert np.array_equal(pipe1.result(pipe1.submit(host[5]))[0].numpy(), want[5])


@pytest.mark.gpu
def test_random_mix_of_sizes_sources_and_collection_lags():
    """Thirty batches through a 2-deep and a 3-deep pipeline without graphs: random sizes (1 .. 24 crops), device or pinned host
    source, results collected with a random lag inside the depth window - every record equals the one-batch-at-a-time result,
    rows beyond the batch's size are zero, the valid-row count travels with the record."""
    from probpose_code_amd.engine import ProbPoseEngine

    flip = S.COCO_FLIP_INDICES
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision="bf16", device="cuda:0")
    rng = np.random.default_rng(3)
    pool = S.synthetic_crops(24, seed=500)
    want = {}
    for depth in (2, 3):
        pipe = StepPipeline(eng, 24, flip, depth=depth, use_graph=False)
        pending = []
        for step in range(30):
            n = int(rng.integers(1, 25))
            x = pool[:n].clone()
            x[0, 0, 0, 0] = step  # batches of equal size still differ
            if n not in want or True:
                ref = pack_records(eng.forward(x.cuda(), True, flip)).cpu().numpy().copy()
            src = x.pin_memory() if rng.random() < 0.5 else x.cuda()
            pending.append((pipe.submit(src), n, ref))
            while len(pending) > int(rng.integers(0, depth)):
                t, m, r = pending.pop(0)
                got = pipe.result(t)[0].numpy()
                assert np.array_equal(got[:m], r), (depth, step)
                assert not got[m:].any()
                assert pipe.gather_of(t).counts == [m]
        for t, m, r in pending:
            assert np.array_equal(pipe.result(t)[0].numpy()[:m], r)

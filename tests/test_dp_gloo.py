"""CPU, world_size 2 and 8, gloo: the data-parallel host logic of texttoaudiogrounding_amd.runner -- every trainable
parameter re-homed into one flat buffer, ONE all-reduce(sum) of the flat gradient, the 1/N mean applied afterwards --
reproduces the single-process gradient of the concatenated batch (equal shard sizes, per-replica mean loss)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from texttoaudiogrounding_amd.runner import FlatParams


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _toy()
        flat = FlatParams(model)
        g = torch.Generator().manual_seed(123)
        torch.set_num_threads(1)
        x, y = torch.randn(16, 12, generator=g), torch.randn(16, 1, generator=g)
        k = 16 // world
        xs, ys = x[rank * k:(rank + 1) * k], y[rank * k:(rank + 1) * k]      # this rank's shard
        flat.zero_grad()
        ((model(xs) - ys) ** 2).mean().backward()                          # per-replica mean
        assert all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in flat.params)   # grads live in the flat buffer
        dist.all_reduce(flat.grad)                                         # the ONE collective of the step
        avg = flat.grad / world
        # both ranks hold identical reduced gradients -> identical clip coefficient without more communication
        gathered = [torch.empty_like(avg) for _ in range(world)]
        dist.all_gather(gathered, avg)
        assert all(torch.equal(gathered[0], g_) for g_ in gathered[1:])
        if rank == 0:
            torch.save(avg, out)
    finally:
        dist.destroy_process_group()


WORLDS = [2, 8]          # 8 = the node BASELINE configs[2] names (8 ranks on this container's 8 cores, one thread each)


@pytest.mark.parametrize("world", WORLDS)
def test_flat_allreduce_matches_single_process(tmp_path, world):
    out = str(tmp_path / "avg.pt")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    avg = torch.load(out)
    model = _toy()
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(123)
    x, y = torch.randn(16, 12, generator=g), torch.randn(16, 1, generator=g)
    flat.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    assert torch.allclose(avg, flat.grad, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# The REAL host logic: StrongRunner's constructor (parameter / buffer broadcast, per-rank dropout seeds), FlatParams over
# the real BiEncoder(Cnn8Rnn + EmbeddingAgg + DotProduct) parameter set, GradBuckets driven through the same
# ready()/flush()/finish() protocol the HIP autograd nodes use -- with the gradients themselves produced by the CPU
# oracle (the HIP kernels cannot run here).  Ragged waveform_len: the exchanged gradient is the SUM over ranks of the
# per-replica-mean gradients (documented deviation from a global mask-count mean).
# ---------------------------------------------------------------------------------------------------------------
S_TOY = 32000


def _real_model(seed):
    from texttoaudiogrounding_amd.models import audio_encoder, audio_text_model, match, text_encoder
    torch.manual_seed(seed)
    return audio_text_model.BiEncoder(audio_encoder.Cnn8Rnn(32000), text_encoder.EmbeddingAgg(5221, 512),
                                      match.DotProduct(), 512)


def _shard(rank):
    from oracle import tag_oracle as O
    return O.synthetic_batch(2, S_TOY, seed=50 + rank, ragged=True)


def _oracle_grads(state, batch):
    from oracle import tag_oracle as O
    buf = ("running_mean", "running_var", "num_batches_tracked", "melspec_extractor")
    st = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not any(b in k for b in buf)
              else v.clone()) for k, v in state.items()}
    loss, _ = O.train_step_loss(st, batch, "dot", "cnn8rnn", True, (0.0, 0.0))
    loss.backward()
    return {k: v.grad for k, v in st.items() if v.is_floating_point() and v.grad is not None}, float(loss)


def _threads(world):
    """Intra-op threads of the oracle, the same in the workers and in the single-process check: the fp32 reduction order of
    the CPU kernels depends on it, and a train-mode BatchNorm at B = 2 turns a last-bit difference into ReLU flips."""
    return 1 if world > 2 else 4


def _real_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from texttoaudiogrounding_amd import ops
        from texttoaudiogrounding_amd.runner import StrongRunner
        torch.set_num_threads(_threads(world))
        model = _real_model(seed=100 + rank)                  # ranks start DIFFERENT on purpose
        model.audio_encoder.bn0.running_mean.fill_(float(rank))
        runner = StrongRunner(model, device="cpu", bucket_bytes=2 << 20)
        assert runner.world == world and runner.rank == rank and ops.SEED_RANK == rank
        # (1) the constructor made the replicas identical: parameters and BatchNorm buffers come from rank 0
        flats = [torch.empty_like(runner.flat.flat) for _ in range(world)]
        dist.all_gather(flats, runner.flat.flat)
        assert all(torch.equal(flats[0], f_) for f_ in flats[1:])
        assert float(model.audio_encoder.bn0.running_mean[0]) == 0.0
        # (2) per-rank dropout seeds differ although both ranks seed torch alike
        torch.manual_seed(0)
        seed = torch.tensor([ops.new_seed()])
        seeds = [torch.empty_like(seed) for _ in range(world)]
        dist.all_gather(seeds, seed)
        assert len({int(s_) for s_ in seeds}) == world
        # (3) gradients of this rank's shard (oracle), delivered in backward order through the sink/bucket protocol
        names = dict((id(p), n) for n, p in model.named_parameters())
        grads, loss = _oracle_grads(model.state_dict(), _shard(rank))
        bk = runner.buckets
        assert len(bk.bounds) >= 3 and bk.bounds[0][0] == 0 and bk.bounds[-1][1] == runner.flat.numel
        runner.flat.zero_grad()
        bk.reset()
        launched_early = 0
        for p in reversed(runner.flat.params):
            p._tag_grad_sink.copy_(grads[names[id(p)]])
            bk.ready([p])
            bk.flush()
            launched_early = max(launched_early, sum(bk.launched))
        assert launched_early == len(bk.bounds)               # every bucket went out from inside "backward"
        bk.finish()
        both = [torch.empty_like(runner.flat.grad) for _ in range(world)]
        dist.all_gather(both, runner.flat.grad)
        assert all(torch.equal(both[0], b_) for b_ in both[1:])   # identical reduced gradients -> identical clip + Adam
        # (4) the aliasing guard
        runner.flat.check()
        model.zero_grad()                                     # set_to_none detaches p.grad from the flat buffer
        with pytest.raises(RuntimeError):
            runner.flat.check()
        if rank == 0:
            torch.save({"sum": runner.flat.grad.clone(), "state": {k: v.clone() for k, v in model.state_dict().items()},
                        "names": [names[id(p)] for p in runner.flat.params]}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_real_runner_host_logic(tmp_path, world):
    out = str(tmp_path / "r.pt")
    mp.spawn(_real_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    # single process: the shards' per-replica-mean gradients, summed
    want = None
    before = torch.get_num_threads()
    torch.set_num_threads(_threads(world))
    try:
        for rank in range(world):
            g, _ = _oracle_grads(got["state"], _shard(rank))
            flat = torch.cat([g[n].reshape(-1) for n in got["names"]])
            want = flat if want is None else want + flat
    finally:
        torch.set_num_threads(before)
    err = (got["sum"] - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-5, err


def _bf16_wire_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from texttoaudiogrounding_amd.runner import GradBuckets
        model = _toy()
        flat = FlatParams(model)
        bk = GradBuckets(flat, bucket_bytes=64, comm_dtype=torch.bfloat16)        # several small buckets
        g = torch.Generator().manual_seed(7 + rank)
        local = torch.randn(flat.numel, generator=g)
        flat.grad.copy_(local)
        bk.reset()
        bk.ready(flat.params)
        bk.flush()
        bk.finish()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert flat.grad.dtype == torch.float32
        if world == 2:                                                           # one addition: its order cannot matter
            want = (both[0].bfloat16() + both[1].bfloat16()).float()             # bf16 payload, bf16 sum, widened back
            assert torch.equal(flat.grad, want)
        exact = torch.stack(both).double().sum(0)
        err = (flat.grad.double() - exact).abs()
        # every rank's cast and every addition of the collective rounds to nearest bf16 (8 significant
        # bits: at most 2^-8 relative per rounding).  Hard bound: N roundings of at most 2^-8 of the magnitudes summed so far
        # <= N * 2^-8 * sum_r |x_r| per element; statistically the roundings add in quadrature and the claim is an rms error
        # <= sqrt(N) * 2^-9 of the rms of the exact sum.
        mag = torch.stack(both).double().abs().sum(0)
        assert bool((err <= world * 2.0 ** -8 * mag).all())
        rel_rms = float(err.pow(2).mean().sqrt() / exact.pow(2).mean().sqrt())
        assert rel_rms <= world ** 0.5 * 2.0 ** -9, rel_rms
        red = [torch.empty_like(flat.grad) for _ in range(world)]
        dist.all_gather(red, flat.grad)
        assert all(torch.equal(red[0], r_) for r_ in red[1:])                    # every rank holds the same reduced gradient
        if rank == 0:
            open(out, "w").write(str(len(bk.bounds)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_bf16_allreduce_payload(tmp_path, world):
    """Opt-in bf16 wire (half the bytes on xGMI; the flat gradient stays fp32) and its error against the sqrt(N) * 2^-9 claim
    of DESIGN.md section 5 at the world sizes 2 and 8."""
    out = str(tmp_path / "n")
    mp.spawn(_bf16_wire_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert int(open(out).read()) >= 2


def test_flat_params_rehoming():
    model = _toy()
    before = [p.detach().clone() for p in model.parameters()]
    flat = FlatParams(model)
    assert flat.numel == sum(p.numel() for p in model.parameters())
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b)
    flat.flat.mul_(2.0)                              # an in-place optimiser update on the flat buffer ...
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), 2 * b)        # ... is visible through every parameter
    model(torch.randn(3, 12)).sum().backward()
    assert flat.grad.abs().sum() > 0
    flat.zero_grad()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in model.parameters())


def test_init_distributed_single_process(monkeypatch):
    from texttoaudiogrounding_amd.runner import init_distributed
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert init_distributed() == (0, 1, 0)

"""RCCL smoke on one GPU: a 1-rank 'nccl' process group must initialise on the box and all-reduce the flat gradient buffer
of StrongRunner in stream order with the HIP kernels (the N > 1 logic itself is covered on CPU by tests/test_dp_gloo.py;
real multi-GPU runs are the driver's)."""
import os
import socket
import warnings

import pytest
import torch
import torch.distributed as dist

from oracle import tag_oracle as O

pytestmark = pytest.mark.gpu


def test_rccl_allreduce_of_flat_gradients(dev):
    from tests.test_gpu_path import build_hip_model
    from texttoaudiogrounding_amd.runner import StrongRunner
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        _rccl_step(dev, port)
    # first-contact hygiene: the communicator is bound to the device at init time and the barrier names it -- torch must not
    # have had to guess ("using the device under current context")
    bad = [str(w.message) for w in caught if "device" in str(w.message).lower() and "barrier" in str(w.message).lower()]
    assert not bad, bad


def _rccl_step(dev, port):
    from tests.test_gpu_path import build_hip_model
    from texttoaudiogrounding_amd.runner import StrongRunner, barrier, comm_environment
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        env = comm_environment()
        assert env["rccl_version"][0].isdigit(), env
        st = O.init_state(seed=1, logit_gain=40.0)
        batch = O.synthetic_batch(2, 32000, seed=3)
        model = build_hip_model(st, "dot", dev).train()
        runner = StrongRunner(model, device=str(dev))
        assert runner.world == 1
        loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        g0 = runner.flat.grad.clone()
        dist.all_reduce(runner.flat.grad)               # what forward_backward does when world > 1
        barrier(dev.index or 0)
        torch.cuda.synchronize()
        assert torch.equal(runner.flat.grad, g0) and torch.isfinite(loss)
        runner.optimizer_step()
        torch.cuda.synchronize()
        assert torch.isfinite(runner.flat.flat).all()
    finally:
        dist.destroy_process_group()


def _one_rank_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))


def test_rccl_buckets_in_flight_beside_cooperative_gru_backward(dev):
    """The ordering DESIGN.md section 5 relies on (functions.py: the fc1 / GRU bucket is announced right after the persistent
    cooperative GRU backward is ENQUEUED): with a real RCCL communicator the bucket collectives are issued from inside
    backward on the communication stream while the spinning GRU workgroups and the side-stream wgrad kernels are in flight.
    Asserted over 3 steps at a size whose GRU runs the cooperative kernels (B = 16, 10 s clips): no exchange timeout / hang,
    gradients bit-identical to the same step with the exchange deferred to the end of backward (overlap_comm=False) and to the
    step without any exchange, every bucket timed on the communication stream, and the bf16 payload = the bf16 rounding of
    the fp32 gradient."""
    from tests.test_gpu_path import build_hip_model
    from texttoaudiogrounding_amd import ops
    from texttoaudiogrounding_amd.runner import StrongRunner
    _one_rank_group()
    try:
        st = O.init_state(seed=4, logit_gain=40.0)
        batch = O.synthetic_batch(16, 320000, seed=8, ragged=True)
        fresh = lambda: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        grads = {}
        for tag, kw in (("none", dict()), ("deferred", dict(force_comm=True, overlap_comm=False)),
                        ("overlap", dict(force_comm=True, overlap_comm=True)),
                        ("overlap_bf16", dict(force_comm=True, overlap_comm=True, grad_comm_dtype=torch.bfloat16))):
            model = build_hip_model(st, "dot", dev).train()
            model.audio_encoder.dropout_p = (0.0, 0.0)
            runner = StrongRunner(model, device=str(dev), bucket_bytes=8 << 20, **kw)
            assert (runner.buckets is not None) == ("force_comm" in kw)
            if runner.buckets is not None:
                assert runner.buckets.stream_wait and len(runner.buckets.bounds) >= 4
                runner.buckets.record = True
            for _ in range(3):
                loss = runner.forward_backward(fresh())
            runner.loss_value(loss)                                  # raises on a GRU exchange timeout
            torch.cuda.synchronize()
            grads[tag] = runner.flat.grad.clone()
            if runner.buckets is not None:
                ts = runner.buckets.timing_summary()
                assert all(v is not None and v >= 0 for v in ts["bucket_ms"]) and ts["exposed_ms_per_step"] is not None
                print(f"{tag}: bucket ms {ts['bucket_ms']}, exposed {ts['exposed_ms_per_step']} ms/step")
        assert torch.equal(grads["none"], grads["deferred"]) and torch.equal(grads["none"], grads["overlap"])
        assert torch.equal(grads["overlap_bf16"], grads["none"].to(torch.bfloat16).float())
    finally:
        dist.destroy_process_group()

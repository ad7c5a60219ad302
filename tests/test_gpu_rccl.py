"""RCCL smoke on one GPU: a 1-rank 'nccl' process group must initialise on the box and all-reduce the flat gradient buffer
of StrongRunner in stream order with the HIP kernels (the N > 1 logic itself is covered on CPU by tests/test_dp_gloo.py;
real multi-GPU runs are the driver's)."""
import os
import socket

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
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        st = O.init_state(seed=1, logit_gain=40.0)
        batch = O.synthetic_batch(2, 32000, seed=3)
        model = build_hip_model(st, "dot", dev).train()
        runner = StrongRunner(model, device=str(dev))
        assert runner.world == 1
        loss = runner.forward_backward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        g0 = runner.flat.grad.clone()
        dist.all_reduce(runner.flat.grad)               # what forward_backward does when world > 1
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(runner.flat.grad, g0) and torch.isfinite(loss)
        runner.optimizer_step()
        torch.cuda.synchronize()
        assert torch.isfinite(runner.flat.flat).all()
    finally:
        dist.destroy_process_group()

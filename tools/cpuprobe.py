import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import tag_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; cat /sys/fs/cgroup/cpu.max 2>/dev/null")
st = O.state_to(O.init_state(seed=0), torch.float32, requires_grad=True)
b = O.synthetic_batch(2, 320000, seed=1234)
for nt in (8, 16, 32, 64):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    ts = []
    for i in range(2):
        t0 = time.perf_counter()
        loss, _ = O.train_step_loss(st, b, "dot", "cnn8rnn", True)
        loss.backward()
        ts.append(time.perf_counter() - t0)
    print(f"threads {nt}: B=2 fwd+bwd {ts} -> {2/ts[-1]:.2f} clips/s", flush=True)

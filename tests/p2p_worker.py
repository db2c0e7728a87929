"""One rank of the two-process peer-exchange test (tests/test_gpu_comm.py starts two of these on the one GPU of the box).
usage: p2p_worker.py <rank> <n_ranks> <dir>   — IPC handles travel as files in <dir>, results as <dir>/result<rank>.npz"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lio_slam_amd as la
from lio_slam_amd import synth
from lio_slam_amd.shard import shard_range

rank, n_ranks, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
c = synth.make_config(1, small=True)
lo, hi = shard_range(len(c["scan"]), rank, n_ranks)
out = {}
for tag, cfg in (("conv", dict(max_iterations=35, eps=1e-2)), ("fixed", dict(max_iterations=6, eps=0.0))):
    n = la.NDT(**cfg)
    n.set_target(c["map"]); n.set_source(c["scan"][lo:hi])
    mine = n.comm_p2p_export()
    tmp = os.path.join(d, f"h_{tag}_{rank}.tmp")
    open(tmp, "wb").write(mine); os.rename(tmp, os.path.join(d, f"h_{tag}_{rank}.bin"))
    handles = []
    for r in range(n_ranks):
        f = os.path.join(d, f"h_{tag}_{r}.bin")
        t0 = time.time()
        while not os.path.exists(f):
            if time.time() - t0 > 60:
                raise SystemExit(f"rank {rank}: no handle from rank {r}")
            time.sleep(0.01)
        handles.append(open(f, "rb").read())
    n.comm_p2p_init(rank, n_ranks, handles)
    assert n.comm_size() == n_ranks
    e = n.eval(c["guess"])
    r1 = n.align(c["guess"])
    r2 = n.align(r1["T"])
    for k, v in (("H", e["H"]), ("b", e["b"]), ("neff", e["n_effective"]), ("T", r1["T"]), ("it", r1["iterations"]), ("conv", r1["converged"]),
                 ("H1", r1["H"]), ("T2", r2["T"])):
        out[f"{tag}_{k}"] = np.asarray(v)
    # both ranks are past their last exchange before either frees its mailbox
    open(os.path.join(d, f"done_{tag}_{rank}"), "w").write("1")
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"done_{tag}_{r}")) for r in range(n_ranks)):
        if time.time() - t0 > 60:
            raise SystemExit(f"rank {rank}: peers did not finish")
        time.sleep(0.01)
    n.comm_destroy(); n.close()
np.savez(os.path.join(d, f"result{rank}.npz"), **out)
print("rank", rank, "ok")
